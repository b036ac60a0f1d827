"""config 5 (PGD-20) as a later configuration of ONE process replayed 3x slower than alone.  Runs bench.py's own
run_config in sequences and prints the event-timed ms per step of every configuration.
argv[1]: comma list from {c3, b512, c2, c2bf, c5}; e.g.  c5   |  c3,c5  |  c3,b512,c2,c2bf,c5"""
import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench

seq = sys.argv[1].split(",")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
pool = bench.make_pool(16 * 1024, dev, seed=1234)
base = {"constraint": "product", "pgd": 0, "pgd_eps": 0.5, "bf16": False, "pre_extracted": False, "no_graph": False, "int16": False}
for name in seq:
    opt = dict(base); batch = 1024; steps, warm = 50, 10
    if name == "b512": batch = 512
    if name in ("c2", "c2bf"): opt["pre_extracted"] = True
    if name == "c2bf": opt["bf16"] = True
    if name == "c5": opt["pgd"] = 20; steps, warm = 20, 5
    dt, ex = bench.run_config(opt, pool, batch, 0, 1, dev, steps, warm, profile=True)
    print(f"{name:5s} ms/step {dt / steps * 1e3:8.3f}  event {ex['event_ms_per_step']:8.3f}  train_graph {ex['train_graph_ms']:8.3f}  mfcc_cus {ex.get('mfcc_cus')} train_cus {ex.get('train_cus')}", flush=True)
