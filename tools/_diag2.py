import sys, time, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import rel_err_rows, d_err, golden_weights
from oracle import posendf_np as onp
from posendf_amd import PoseNDF, amass_config, synth
def mk(prec, sd):
    cfg = amass_config('lrelu','cuda:0'); cfg['engine']={'precision':prec}
    net = PoseNDF(cfg); net.load_state_dict({k: torch.from_numpy(v) for k,v in sd.items()}); return net
for regime in ('mixed','live'):
    sd = golden_weights(regime)
    nets = {p: mk(p, sd) for p in ('fp32','f16x3')}
    qn = synth.make_poses(2048, seed=3, signed=(regime=='mixed'))
    d64, g64 = onp.forward_grad(qn, sd, 'lrelu', dtype=np.float64)
    d32, g32 = onp.forward_grad(qn, sd, 'lrelu')
    print(regime, 'oracle fp32 vs fp64: d', d_err(d32,d64), 'dq median', np.median(rel_err_rows(g32,g64)), 'p99', np.percentile(rel_err_rows(g32,g64),99))
    for p, net in nets.items():
        q = torch.from_numpy(qn).cuda().requires_grad_(True)
        d = net(q, train=False)['dist_pred']; (g,) = torch.autograd.grad(d.sum(), q)
        e = rel_err_rows(g.cpu().numpy(), g64)
        print(' ', p, 'd err', d_err(d.detach().cpu().numpy(), d64), 'dq median', np.median(e), 'p99', np.percentile(e,99), 'frac>1e-4', (e>1e-4).mean())
        for steps in (10, 100):
            qp,_ = net.project(torch.from_numpy(qn[:512]), steps=steps)
            q64,_ = onp.project(qn[:512], sd, steps=steps, dtype=np.float64)
            q32,_ = onp.project(qn[:512], sd, steps=steps)
            em, er = rel_err_rows(qp.cpu().numpy(), q64), rel_err_rows(q32, q64)
            print('    project', steps, 'median', np.median(em), 'frac>1e-4', (em>1e-4).mean(), '(oracle fp32:', np.median(er), (er>1e-4).mean(), ')')
sd = synth.make_weights(0,2.0,0.1)
for p in ('fp32','f16x3'):
    net = mk(p, sd)
    qt = torch.from_numpy(synth.make_poses(65536, seed=1)).cuda()
    net.project(qt, steps=2); torch.cuda.synchronize()
    for steps in (10, 100):
        t0=time.perf_counter(); net.project(qt, steps=steps); torch.cuda.synchronize(); dt=time.perf_counter()-t0
        print(p, 'B=65536 steps', steps, '%.2f ms'%(dt*1e3), '%.0f poses/s'%(65536/dt), '%.1f TFLOP/s algorithmic'%(65536*steps*5450416/dt/1e12))
