import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench, ddsp_piano_amd as dp
dev = torch.device('cuda', 0)
for B in (1, 4):
    feats, base = bench.make_features(B, 16, 750, 128, 96, 1, 72000, dev, 3)
    pg = bench.build_group(dp, 16, 24000)
    for _ in range(3): y = pg(feats)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): y = pg(feats)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): pg(feats)
    torch.cuda.current_stream().wait_stream(s)
    try:
        with torch.cuda.graph(g):
            yg = pg(feats)
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 50
        print(f'B={B}: eager {eager*1e3:.3f} ms, graph replay {graph*1e3:.3f} ms, finite {bool(torch.isfinite(yg).all())}, rms {float(yg.pow(2).mean().sqrt()):.4f}')
    except Exception as e:
        print('capture failed:', repr(e)[:500])
