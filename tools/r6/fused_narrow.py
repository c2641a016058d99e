import sys, numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import test_depth as td, conftest as cf
from oracle import posendf_np as onp
from posendf_amd import PoseNDF, synth
nets = [[256,512,1024,512,256,1], [256,512,1024,1,256,64], [1,512,1024,512,256,64], [256,2,1024,512,2,64], [200,300,1,300,200,50]]
for hidden in nets:
  for act in ("softplus", "lrelu"):
    dims = (126, *hidden, 1)
    try: sd = td.live_weights(dims, act)
    except AssertionError: print(hidden, act, "dead"); continue
    q_np = np.concatenate([synth.make_poses(100, seed=61), synth.make_poses(100, seed=62, signed=True)])
    sig_d, sig_g, d64, g64 = cf.fp32_noise(q_np, sd, act)
    for prec in ("fp32", "f16x3"):
        cfg = td.config_for(hidden, act, True, "cuda:0"); cfg["engine"] = {"precision": prec}
        net = PoseNDF(cfg); net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net.eval()
        q = torch.from_numpy(q_np).cuda().requires_grad_(True)
        d = net(q, train=False)["dist_pred"]; (dq,) = torch.autograd.grad(d.sum(), q)
        e = cf.rel_err_rows(dq.cpu().numpy(), g64)
        bad = e > 8 * sig_g + 8e-6
        print(hidden, act, prec, net._engine_for(q.device).kernel_name(), "median %.1e max %.1e over-gate %d (>1e-2: %d)" % (np.median(e), e.max(), bad.sum(), (e > 1e-2).sum()), flush=True)
