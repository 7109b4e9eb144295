import torch
a, b = torch.load('gpurun_out/sw_base.pt'), torch.load('gpurun_out/sw_new.pt')
for k in a:
    d = (a[k] - b[k]).abs().max() / a[k].abs().max()
    print('%-10s old-vs-new rel %.2e   new vs its own recompute %.2e   old vs its recompute %.2e' % (
        k, float(d), float((b[k] - b['recompute']).abs().max() / b['recompute'].abs().max()) if b[k].shape == b['recompute'].shape else -1,
        float((a[k] - a['recompute']).abs().max() / a['recompute'].abs().max()) if a[k].shape == a['recompute'].shape else -1))
