import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench, ddsp_piano_amd as dp
from ddsp_piano_amd import core
dev = torch.device('cuda', 0)
B, P, T, H, K, S, L, sr = 1, 16, 750, 128, 96, 1, 72000, 24000
feats, base = bench.make_features(B, P, T, H, K, S, L, dev, 3)
add = dp.MultiInharmonic(frame_rate=250, sample_rate=sr, inference=True)
nz = dp.DynamicSizeFilteredNoise(frame_rate=250, sample_rate=sr)
rv = dp.Reverb()
x = torch.randn(B, 72000, device=dev)
zn = torch.rand(B, 72000, device=dev) * 2 - 1
cap_stream = torch.cuda.Stream()
def try_capture(name, fn, mode='global'):
    cap_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap_stream):
        for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=cap_stream, capture_error_mode=mode):
            y = fn()
        g.replay(); torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t0) / 50
        t0 = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize()
        te = (time.perf_counter() - t0) / 50
        print(name, 'captured OK; replay %.3f ms, eager %.3f ms' % (tg * 1e3, te * 1e3))
    except Exception as e:
        print(name, 'FAILED', repr(e)[:120])
        torch.cuda.synchronize()
try_capture('controls', lambda: add.get_controls(feats['amplitudes_0'], feats['harmonic_distribution_0'], feats['inharm_coef_0'], feats['f0_hz_0']))
c = add.get_controls(feats['amplitudes_0'], feats['harmonic_distribution_0'], feats['inharm_coef_0'], feats['f0_hz_0'])
try_capture('additive', lambda: add.get_signal(**c))
try_capture('noise explicit', lambda: nz.get_signal(nz.get_controls(feats['magnitudes_0'])['magnitudes'], noise=zn))
try_capture('noise philox', lambda: nz(feats['magnitudes_0']))

pg = bench.build_group(dp, P, sr)
try_capture('reverb', lambda: rv(x, feats['reverb_ir']))
try_capture('group', lambda: pg(feats))
try_capture('group dict', lambda: pg(feats, return_outputs_dict=True))
