import torch, numpy as np
a, b = torch.load('gpurun_out/eng_base.pt'), torch.load('gpurun_out/eng_new.pt')
M, KB = 8, 3
for key in ('act', 'act_b'):
    A, B = a[key], b[key]
    Lm, rows, cols = A.shape
    A = A.reshape(Lm, cols // 16, rows, 16); B = B.reshape(Lm, cols // 16, rows, 16)
    for st in (0, 1):
        wa = A[:, :, rows - 4 + 2 * st: rows - 2 + 2 * st].contiguous().numpy().view(np.uint32).reshape(Lm, cols // 16, 64)
        wb = B[:, :, rows - 4 + 2 * st: rows - 2 + 2 * st].contiguous().numpy().view(np.uint32).reshape(Lm, cols // 16, 64)
        tot = 0; live = 0
        for j in range(M - 1):
            for r in range(KB):
                oa = (wa >> (4 * j + r)) & 1
                ob = (wb >> (KB * (M - 1) - 1 - (KB * j + r))) & 1
                mm = (oa != ob)
                tot += int(mm.sum())
                # live rows: r < 2 all lanes, r == 2 only lane groups g < 2 (lanes 0..31)
                if r < 2: live += int(mm.sum())
                else: live += int(mm[..., :32].sum())
        print(key, 'stage', st, 'mask mismatches', tot, 'in live rows', live, ' upper bits of new words nonzero:', int(((wb >> 21) != 0).sum()))
