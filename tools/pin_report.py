"""Table of a finished pin run (tools/dr_pin.sh, tools/robust_plr_pin.sh) against the reference's result file: final-checkpoint
evaluation (eval.py, 100 episodes) vs the reference's per-seed values, and the run's in-training evaluations per 50 M-step window.
Needs /root/reference (build container only); the outputs under profiles/ are data + this table.
  usage: python tools/pin_report.py <reference results csv> <logs.csv> <eval.json>   (markdown on stdout)"""
import csv, json, sys
import numpy as np
ref_csv, logs, ev_json = sys.argv[1:4]
ref = {r[0]: [float(x) for x in r[1:]] for r in csv.reader(open(ref_csv)) if r and r[0] != "metric"}
ours = json.load(open(ev_json))
rows = list(csv.DictReader(open(logs)))
print(f"| metric | this run (final checkpoint, 100 episodes) | reference min | mean | max ({len(next(iter(ref.values())))} seeds) | |")
print("|---|---|---|---|---|---|")
for k, v in ref.items():
    if k in ours:
        a = np.array(v)
        where = "inside" if a.min() <= ours[k] <= a.max() else ("above" if ours[k] > a.max() else "below")
        print(f"| {k} | {ours[k]:.2f} | {a.min():.2f} | {a.mean():.2f} | {a.max():.2f} | {where} |")
names = [k[len("solved_rate:"):] for k in rows[0] if k.startswith("solved_rate:")]
ev = [r for r in rows if names and r.get("solved_rate:" + names[0]) not in (None, "")]
s = np.array([float(r["steps"]) for r in ev])
print("\nIn-training evaluations of the same run (10 episodes each), mean solved rate per 50 M-step window:\n")
print("| env | 0-50 M | 50-100 M | 100-150 M | 150-200 M | 200-250 M | reference mean at 250 M |")
print("|---|---|---|---|---|---|---|")
for name in names:
    x = np.array([float(r["solved_rate:" + name]) for r in ev])
    cells = []
    for lo, hi in ((0, 50e6), (50e6, 100e6), (100e6, 150e6), (150e6, 200e6), (200e6, 251e6)):
        m = (s >= lo) & (s < hi)
        cells.append(f"{x[m].mean():.2f} (n={m.sum()})" if m.any() else "-")
    print(f"| {name} | " + " | ".join(cells) + f" | {np.mean(ref['solved_rate:' + name]):.2f} |")
last = rows[-200:]
f = lambda k: np.mean([float(r[k]) for r in last if r.get(k) not in (None, "")]) if any(r.get(k) not in (None, "") for r in last) else float("nan")
print(f"\nLast 200 log rows: mean_agent_return {f('mean_agent_return'):.2f}, total_student_grad_updates {float(rows[-1]['total_student_grad_updates']):.0f}, "
      f"steps {float(rows[-1]['steps']):.0f}, sps (median) {np.median([float(r['sps']) for r in rows[5:]]):.0f}")
