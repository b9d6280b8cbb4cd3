"""Same-box A/B of the weight pack plan on the eagerly issued step: PLAN=0|1 python tools/plan_ab.py [workload]"""
import argparse, os, sys, time, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/speech-separation_amd")
import bench
from dtt_amd import ops
from dtt_amd.dist import GradBucketReducer
plan = os.environ.get("PLAN","0")=="1"
args = argparse.Namespace(batch=None, precision="bf16", graph="0")
w = bench.build_workload(sys.argv[1] if len(sys.argv)>1 else "xattn60m", args, torch.device("cuda",0), 0, 1, GradBucketReducer)
ops.weight_plan(plan)
for i in range(5): w["step"](i)
torch.cuda.synchronize()
t0=time.perf_counter()
n=30
for i in range(n): l=w["step"](i)
torch.cuda.synchronize()
print("plan",plan,"ms/step",(time.perf_counter()-t0)/n*1e3, "loss", float(l.detach()))
