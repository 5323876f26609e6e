import os, sys, time, torch, argparse
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench, ddsp_piano_amd as dp
dev = torch.device('cuda', 0)
sys.argv = ['bench.py']
args = bench.parse()
feats, base = bench.make_features(64, 16, 750, 128, 96, 1, 72000, dev, 20240)
r = bench.measure_roofline(dp, base, args, 750, 96, dev); print('cold      ', r['ms_per_launch'], r['frac'])
pg = bench.build_group(dp, 16, 24000)
for _ in range(300): pg(feats)
torch.cuda.synchronize()
r = bench.measure_roofline(dp, base, args, 750, 96, dev); print('after 300 steps', r['ms_per_launch'], r['frac'])
time.sleep(3)
r = bench.measure_roofline(dp, base, args, 750, 96, dev); print('after 3 s idle ', r['ms_per_launch'], r['frac'])
