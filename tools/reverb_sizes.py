import os, sys, torch, subprocess, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
if len(sys.argv) > 1:
    import ddsp_piano_amd as dp
    B, N, L = 64, 72000, 72000
    x = torch.randn(B, N, device='cuda'); ir = torch.randn(B, L, device='cuda') * 0.01
    rv = dp.Reverb(name='reverb')
    for _ in range(3): y = rv(x, ir)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = rv(x, ir)
    e1.record(); e1.synchronize()
    print(sys.argv[1], 'ms per call', e0.elapsed_time(e1) / 20, float(y.abs().mean()))
else:
    for sz in [262144, 144000, 147456, 163840, 196608, 150000, 153600, 160000, 144384, 145800, 149760]:
        env = dict(os.environ); env['DDSPP_FFT_SIZE'] = str(sz)
        r = subprocess.run([sys.executable, __file__, str(sz)], env=env, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
