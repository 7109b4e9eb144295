import torch, sys, numpy as np
a, b = torch.load('gpurun_out/act_base.pt'), torch.load('gpurun_out/act_new.pt')
M, KB = 8, 3
for xo in (0, 1):
    A, B = a['act%d' % xo], b['act%d' % xo]
    print('x_only', xo, 'u max diff', float((a['u%d' % xo] - b['u%d' % xo]).abs().max()))
    Lm, rows, cols = A.shape
    A = A.reshape(Lm, cols // 16, rows, 16); B = B.reshape(Lm, cols // 16, rows, 16)
    data = slice(0, rows - 4)
    da = (A[:, :, data] - B[:, :, data]).abs()
    both_nan = torch.isnan(A[:, :, data]) & torch.isnan(B[:, :, data])
    one_nan = torch.isnan(A[:, :, data]) ^ torch.isnan(B[:, :, data])
    da[both_nan] = 0
    print('   data rows: max |diff|', float(da[~one_nan].max()), ' slots written by only one of them:', int(one_nan.sum()), ' rows', sorted(set(torch.nonzero(one_nan)[:, 2].tolist()))[:20])
    bad = torch.nonzero(da > 1e-9)
    print('   rows with diff > 1e-9:', sorted(set(bad[:, 2].tolist()))[:40], 'count', len(bad))
    for st in (0, 1):
        wa = A[:, :, rows - 4 + 2 * st: rows - 2 + 2 * st].contiguous().numpy().view(np.uint32).reshape(Lm, cols // 16, 64)
        wb = B[:, :, rows - 4 + 2 * st: rows - 2 + 2 * st].contiguous().numpy().view(np.uint32).reshape(Lm, cols // 16, 64)
        mism = 0
        for j in range(M - 1):
            for r in range(KB):
                oa = (wa >> (4 * j + r)) & 1
                ob = (wb >> (KB * (M - 1) - 1 - (KB * j + r))) & 1
                mism += int((oa != ob).sum())
        print('   stage', st, 'mask mismatches', mism, 'of', wa.size * 21)
