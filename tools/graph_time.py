import importlib, sys, time, torch
sys.path.insert(0, "/root/repo")
sed = importlib.import_module("soundeventdetection-pytorch_amd")
cfg = [(32, 2), (64, 2), (128, 2), (128, 1)]
for (B, T) in ((4, 30), (32, 182), (32, 6001)):
    for graph in (False, True):
        torch.manual_seed(0)
        m = sed.Cnn_AvgPooling(1, cfg, precision="bf16").cuda()
        tr = sed.FusedTrainer(m, lr=1e-6, graph=graph)
        x = torch.randn(B, 1, T, 64, device="cuda"); y = (torch.rand(B, T, 1, device="cuda") < 0.04).float()
        for _ in range(5): tr.train_step(x, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 50 if T < 1000 else 20
        for _ in range(n): tr.train_step(x, y)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"B={B} T={T} graph={graph}: {dt*1e3:.3f} ms/step  {B/dt:.0f} clips/s", flush=True)
