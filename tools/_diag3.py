import sys, time, numpy as np, torch
sys.path.insert(0,'/root/repo')
from posendf_amd import PoseNDF, amass_config, synth
sd = synth.make_weights(0,2.0,0.1)
cfg = amass_config('lrelu','cuda:0'); cfg['engine']={'precision': sys.argv[1] if len(sys.argv)>1 else 'f16x3'}
net = PoseNDF(cfg); net.load_state_dict({k: torch.from_numpy(v) for k,v in sd.items()})
qt = torch.from_numpy(synth.make_poses(65536, seed=1)).cuda()
net.project(qt, steps=2); torch.cuda.synchronize()
t0=time.perf_counter(); net.project(qt, steps=50); torch.cuda.synchronize(); dt=time.perf_counter()-t0
print(cfg['engine'], 'steps 50: %.2f ms  -> %.0f poses/s at 100 steps'%(dt*1e3, 65536/(2*dt)))
