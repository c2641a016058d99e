import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
from oracle import posendf_np as onp
from posendf_amd import PoseNDF, amass_config, synth
sd = synth.make_weights(0, 2.5, 0.05)
net = PoseNDF(amass_config('lrelu', 'cuda:0'))
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
qn = synth.make_poses(64, seed=7, signed=True)
q = torch.from_numpy(qn).cuda().requires_grad_(True)
d = net(q, train=False)['dist_pred']
(g,) = torch.autograd.grad(d.sum(), q)
dbg = {}
do, go = onp.forward_grad(qn, sd, 'lrelu', debug=dbg)
g = g.cpu().numpy()
err = np.abs(g - go)
print('per-joint max err:', np.array2string(err.max(axis=(0,2)), precision=2))
print('per-component max err:', err.max(axis=(0,1)))
print('per-pose max err (first 32):', np.array2string(err.max(axis=(1,2))[:32], precision=2))
print('scale', np.abs(go).max())
# is it the normalisation backward only? compare against gn/denom-type quantities
gn = dbg['gn']
print('corr with gn/denom?', np.abs(g - gn/ np.maximum(np.sqrt((qn*qn).sum(1,keepdims=True)),1e-12)).max())
