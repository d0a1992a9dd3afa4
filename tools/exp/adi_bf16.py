import sys, os, time, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
from bench_cfg5 import DeepCubeStandIn
from bench_adi_pipeline import run
dev = torch.device('cuda')
m = DeepCubeStandIn().to(dev).eval().to(torch.bfloat16)
print(json.dumps(run(sizes=((200,30),(20000,30),(100000,30)), reps=3, model=m, dev=dev)))
