"""CPU: log files in the reference's layout (util/filewriter.py:132-229): header comment, growing fields.csv,
_tick/_time columns, resume tick."""
import csv
import json
import os


def test_filewriter_layout_and_resume(tmp_path):
    from util.filewriter import FileWriter
    w = FileWriter(xpid="run", xp_args={"lr": 1e-4}, rootdir=str(tmp_path))
    w.log({"steps": 10, "sps": 1.5})
    w.log({"steps": 20, "sps": 2.5, "mean_agent_return": 0.1})
    w.close()
    base = tmp_path / "run"
    assert os.path.islink(tmp_path / "latest")
    lines = open(base / "logs.csv").read().splitlines()
    assert lines[0] == "# _tick,_time,steps,sps" and lines[1].startswith("0,") and lines[2].startswith("1,")
    fields = list(csv.reader(open(base / "fields.csv")))
    assert fields[0] == ["_tick", "_time", "steps", "sps"] and fields[-1][-1] == "mean_agent_return"
    meta = json.load(open(base / "meta.json"))
    assert meta["args"]["lr"] == 1e-4 and meta["successful"] is True and meta["xpid"] == "run"
    w2 = FileWriter(xpid="run", xp_args={}, rootdir=str(tmp_path))
    assert w2.latest_tick() == 1
    w2.log({"steps": 30})
    w2.close()
    assert open(base / "logs.csv").read().splitlines()[-1].startswith("2,")
