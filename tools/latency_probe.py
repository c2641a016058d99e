import sys, time, torch
sys.path.insert(0, '/root/repo')
from posendf_amd import PoseNDF, amass_config, synth
cfg = amass_config("lrelu", "cuda:0")
net = PoseNDF(cfg); net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(0, 2.0, 0.1).items()}); net.eval()
for B in (256, 4096):
    q = torch.from_numpy(synth.make_poses(B, seed=1)).cuda()
    for _ in range(20): net(q, train=False)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200): d = net(q, train=False)["dist_pred"]
    torch.cuda.synchronize(); t1 = (time.perf_counter() - t) / 200 * 1e6
    qg = q.clone().requires_grad_(True)
    for _ in range(20): net(qg, train=False)["dist_pred"].mean().backward()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200):
        qg.grad = None
        net(qg, train=False)["dist_pred"].mean().backward()
    torch.cuda.synchronize(); t2 = (time.perf_counter() - t) / 200 * 1e6
    eng = net._engine_for(q.device); dd = torch.empty(B, device="cuda"); st = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200): eng.forward(q.data_ptr(), dd.data_ptr(), B, st)
    torch.cuda.synchronize(); t3 = (time.perf_counter() - t) / 200 * 1e6
    print(f"B={B}: forward {t1:.0f} us/call, forward+backward {t2:.0f} us/call, raw C-ABI forward {t3:.0f} us/call")
