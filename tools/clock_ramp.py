import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench, ddsp_piano_amd as dp
dev = torch.device('cuda', 0)
feats, base = bench.make_features(64, 16, 750, 128, 96, 1, 72000, dev, 20240)
pg = bench.build_group(dp, 16, 24000)
t00 = time.perf_counter()
for w in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): pg(feats)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'window {w:2d} at {time.perf_counter()-t00:5.2f} s: {dt/20*1e3:.3f} ms/step')
