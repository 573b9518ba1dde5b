"""Experiment log files in the reference's layout (reference: util/filewriter.py:65-279), so its plotting /
resume tooling reads runs of this engine: <rootdir>/<xpid>/{logs.csv, fields.csv, meta.json, out.log},
a `latest` symlink, `_tick` / `_time` columns, the `# field,names` comment before the first row, a new row in
fields.csv whenever the set of logged keys grows, and `latest_tick()` for preemption resume (train.py:165-166).
"""
import copy
import csv
import datetime
import json
import os
import time


class FileWriter:
    def __init__(self, xpid=None, xp_args=None, rootdir="~/logs", symlink_to_latest=True, seeds=None):
        self.xpid = xpid or f"{os.getpid()}_{int(time.time())}"
        self._tick = 0
        self.metadata = {"date_start": datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S.%f"), "date_end": None,
                         "successful": False, "args": copy.deepcopy(xp_args or {}), "xpid": self.xpid,
                         "env": {k: v for k, v in os.environ.items() if k.startswith(("SLURM", "ROCM", "HIP", "HSA"))}}
        rootdir = os.path.expandvars(os.path.expanduser(rootdir))
        self.basepath = os.path.join(rootdir, self.xpid)
        os.makedirs(self.basepath, exist_ok=True)
        if symlink_to_latest:
            link = os.path.join(rootdir, "latest")
            try:
                if os.path.islink(link):
                    os.remove(link)
                if not os.path.exists(link):
                    os.symlink(self.basepath, link)
            except OSError:
                pass
        self.paths = {k: os.path.join(self.basepath, v) for k, v in
                      dict(msg="out.log", logs="logs.csv", fields="fields.csv", meta="meta.json").items()}
        self._save_metadata()
        self.fieldnames = ["_tick", "_time"]
        if os.path.exists(self.paths["logs"]):                    # resume: reuse the last field list and tick
            with open(self.paths["fields"]) as f:
                rows = list(csv.reader(f))
                if rows:
                    self.fieldnames = rows[-1]
            self._tick = self.latest_tick() + 1
        self._fieldfile = open(self.paths["fields"], "a")
        self._fieldwriter = csv.writer(self._fieldfile)
        self._logfile = open(self.paths["logs"], "a")
        self._logwriter = csv.DictWriter(self._logfile, fieldnames=self.fieldnames)
        self._msgfile = open(self.paths["msg"], "a")

    def log(self, to_log, tick=None, verbose=False):
        if tick is not None:
            raise NotImplementedError
        to_log["_tick"] = self._tick
        self._tick += 1
        to_log["_time"] = time.time()
        grew = False
        for k in to_log:
            if k not in self.fieldnames:
                self.fieldnames.append(k)
                grew = True
        if grew:
            self._fieldwriter.writerow(self.fieldnames)
            self._fieldfile.flush()
        if to_log["_tick"] == 0:
            self._logfile.write("# %s\n" % ",".join(self.fieldnames))
        if verbose:
            msg = "LOG | " + ", ".join(f"{k}: {to_log[k]}" for k in sorted(to_log))
            print(msg, flush=True)
            self._msgfile.write(msg + "\n")
            self._msgfile.flush()
        self._logwriter.writerow(to_log)
        self._logfile.flush()

    def latest_tick(self):
        last = -1
        with open(self.paths["logs"]) as f:
            for row in csv.reader(f):
                if row and not row[0].startswith("#"):
                    try:
                        last = int(row[0])
                    except ValueError:
                        pass
        return last

    def close(self, successful=True):
        self.metadata["date_end"] = datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S.%f")
        self.metadata["successful"] = successful
        self._save_metadata()
        for f in (self._logfile, self._fieldfile, self._msgfile):
            f.close()

    def _save_metadata(self):
        with open(self.paths["meta"], "w") as f:
            json.dump(self.metadata, f, indent=4, sort_keys=True, default=str)
