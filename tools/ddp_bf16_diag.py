#!/usr/bin/env python3
"""Per-parameter deviation of the 2-rank SyncBN gradient from the single-process full-batch gradient (GPU box; bf16)."""
import importlib
import os
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W = os.path.join(ROOT, "tests", "ddp_gpu_worker.py")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
env = dict(os.environ, SED_TEST_PRECISION=prec, HSA_ENABLE_IPC_MODE_LEGACY="0")
d = tempfile.mkdtemp()


def run(world, mode, out, port):
    ps = [subprocess.Popen([sys.executable, W, str(r), str(world), str(port), mode, out], env=env, cwd=ROOT) for r in range(world)]
    for p in ps:
        assert p.wait(timeout=400) == 0


run(1, "shard", d + "/full.pt", 29611)
run(2, "sync", d + "/sync.pt", 29612)
full, r0 = torch.load(d + "/full.pt"), torch.load(d + "/sync.pt.r0")
sed = importlib.import_module("soundeventdetection-pytorch_amd")
m = sed.Cnn_AvgPooling(1, [(32, 2), (64, 2), (128, 2), (128, 1)])
flat = sed.train.FlatParams(m)
gs, gf = r0["g"].double(), full["g"].double()
print("total rel", float((gs - gf).norm() / gf.norm()))
# generic: walk the flat views
for n in flat.names:
    v = flat.G[n]
    o = v.storage_offset()
    k = v.numel()
    a, b = gs[o:o + k], gf[o:o + k]
    print(f"{n:40s} |full| {float(b.norm()):10.4e}  rel {float((a - b).norm() / (b.norm() + 1e-30)):8.4f}")
print("logits max diff", float((torch.cat([r0['logits'], torch.load(d + '/sync.pt.r1')['logits']], 0) - full['logits']).abs().max()))
