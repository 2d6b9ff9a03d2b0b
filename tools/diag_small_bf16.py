"""diagnostic: bf16 engine vs the bf16-storage oracle on the small parity case, per parameter, for the block-0 / statistics modes"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cnn_oracle_bf16 as OB
MAIN_CFG = [(32, 2), (64, 2), (128, 2), (128, 1)]
sed = importlib.import_module("soundeventdetection-pytorch_amd")
B, Tn = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 256
AFF = len(sys.argv) > 3 and sys.argv[3] == "aff"
for c1, ps in (("1", "p"),):
    os.environ["SED_C1_MODE"], os.environ["SED_POOL_STATS"] = c1, ps
    torch.manual_seed(5)
    model = sed.Cnn_AvgPooling(1, MAIN_CFG, precision="bf16")
    if AFF:
        with torch.no_grad():
            for blk in model.conv_blocks:
                for bn in (blk.bn1, blk.bn2):
                    bn.weight.uniform_(0.7, 1.3)
                    bn.bias.uniform_(-0.2, 0.2)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = torch.randn(B, 1, Tn, 64)
    y = (torch.rand(B, Tn, 1) > 0.8).float()
    MODE = sys.argv[4] if len(sys.argv) > 4 else "autograd"
    if MODE.startswith("clip"):
        g = torch.Generator().manual_seed(77)
        x = torch.randn(B, 1, Tn, 64, generator=g)
        y = torch.zeros(B, Tn, 1)
        for b in range(B):
            for s0 in torch.randint(0, Tn - 80, (6,), generator=g).tolist():
                y[b, s0:s0 + 40] = 1.0
                x[b, 0, s0:s0 + 40] += 1.5
    model.cuda().train()
    if MODE.endswith("trainer"):
        tr = sed.FusedTrainer(model, lr=1e-3, recall_factor=5.0)
        tr.forward_backward(x.cuda(), y.cuda())
        plan = next(iter(model.engine._plans.values()))
        out = model.engine.interpolate(plan)
        class _P:  # named_parameters stand-in
            pass
        grads_dev = {n: tr.flat.G[n] for n in tr.flat.names}
    else:
        out = model(x.cuda())
        loss = sed.WeightedBCE(5, True)(out, y.cuda())
        loss.backward()
        grads_dev = {n: p.grad for n, p in model.named_parameters()}
    plan = next(iter(model.engine._plans.values()))
    _, logits_b, grads_b, _ = OB.train_step_grads_bf16(x, y, sd, MAIN_CFG, 5.0, c1_mode=bool(plan.c1_mode))
    res = {}
    for n, gd in grads_dev.items():
        a, b = gd.double().cpu().flatten(), grads_b[n].double().flatten()
        res[n] = (round(float((a @ b) / (a.norm() * b.norm() + 1e-30)), 5), round(float(a.norm() / b.norm()), 4))
    lo = logits_b.double(); print("logits rel", float((out.double().cpu() - lo).norm() / lo.norm()))
    print(f"C1={c1} pool={ps} c1_mode={plan.c1_mode}", "min cos", min(v[0] for v in res.values()), {k.replace("conv_blocks.", "b"): v for k, v in list(res.items())[:4]})
