import torch
a, b = torch.load('gpurun_out/eng_base.pt'), torch.load('gpurun_out/eng_new.pt')
for k in a:
    x, y = a[k], b[k]
    m = torch.isfinite(x) & torch.isfinite(y)
    print('%-8s rel diff %.3e  (nan mismatch %d)' % (k, float((x[m] - y[m]).abs().max() / x[m].abs().max()), int((torch.isnan(x) ^ torch.isnan(y)).sum())))
