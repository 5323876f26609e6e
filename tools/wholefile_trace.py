import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench, ddsp_piano_amd as dp
dev = torch.device('cuda', 0)
feats, base = bench.make_features(1, 16, 34000, 128, 96, 1, 48000, dev, 11)
pg = bench.build_group(dp, 16, 24000)
for _ in range(10): y = pg(feats)
torch.cuda.synchronize()
