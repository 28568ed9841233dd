"""Per-kernel durations of the pooled rules pipeline at 32,768 boards (S-mid): run under
`rocprofv3 --kernel-trace` and split the trace by grid size -- fused (mask + planes), mask only,
planes only.  Prints nothing useful by itself."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from alphazero_quoridor_amd import rules
from movegen_bench import position_set

dev = torch.device("cuda:0")
n = 32768
db = position_set("S-mid", n, dev)
mask = torch.empty((n, 5), dtype=torch.int32, device=dev)
planes = torch.empty((n, 26, 9, 9), dtype=torch.float32, device=dev)
for _ in range(3):
    rules.movegen_encode(db, mask, planes)
torch.cuda.synchronize()
for _ in range(30):
    rules.movegen_encode(db, mask, planes)
    torch.cuda.synchronize()
for _ in range(30):
    rules.movegen(db)
    torch.cuda.synchronize()
for _ in range(30):
    rules.encode(db, planes)
    torch.cuda.synchronize()
