"""dcd_mg_step alone at large B (bench.py's roofline_large_b leg), for same-box A/B runs of the wide step kernel."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dcd-isaac_amd")]
import torch
import bench
sizes = tuple(int(x) for x in sys.argv[1:]) or (65536, 262144, 1048576)
torch.cuda.set_device(0)
for r in bench.sweep_large_b(torch.device("cuda", 0), "u8", sizes=sizes, iters=50):
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}), flush=True)
