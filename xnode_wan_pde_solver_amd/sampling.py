"""Monte-Carlo sampling of space-time paths: domain shapes and the loader, with the reference's call surface and its
exact RNG consumption (so that a seeded run draws the same points; pinned by tests/golden).

Mirrors (file:line into the reference):
    Hypercube      src/dataset.py:232-290
    Comb_loader    src/dataset.py:293-322
    fillt          src/dataset.py:13-32
Domain protocol (src/dataset.py:34-45): interior(N_r), boundary(N_b), func_w(x), bound_pad(x), V().
Data layout: [N, L, 1+d], time in channel 0; on the cube every path is one spatial point repeated over the shared,
sorted random time grid.

Sampling stays on the host RNG (torch CPU generator) -- it is O(N d) work per outer iteration and must be
draw-for-draw identical to the reference for "same seeds" parity; the engine uploads only what the kernels need.
"""
import torch
from torch.utils.data import Dataset


def _time_grid(T0, T, N_t):
    grid, _ = torch.sort(torch.Tensor(N_t).uniform_(T0, T), 0)
    grid[0], grid[-1] = T0, T
    return grid


def _paths(times, x):
    """[n, d] spatial points -> [n, L, 1+d] vertical paths over the time grid."""
    n, d = x.shape
    L = times.shape[0]
    out = torch.empty(n, L, d + 1)
    out[:, :, 0] = times.view(1, L)
    out[:, :, 1:] = x.view(n, 1, d)
    return out


def fillt(inputs, T, T0, min_steps=5):
    """Densify a time vector so that no gap exceeds (T - T0) / min_steps; returns (index map, filled times)
    (reference src/dataset.py:13-32: evaluation-time helper reached through bound_pad)."""
    times = inputs
    step = (T - T0) / min_steps
    gaps = torch.cat((torch.tensor(step / 2).view(1).to(times.device), torch.abs(times[:-1] - times[1:])), 0)
    big = torch.nonzero(gaps > step).squeeze(1)
    big = torch.cat((big, torch.tensor([times.shape[0]]).to(times.device)), 0).int()
    index = torch.arange(times.shape[0]).to(times.device)
    out = times[0].view(1)
    for k in range(big.shape[0] - 1):
        stop = times[big[k]].item()
        n = round((stop - 2 * step - out[-1].item()) / step) + 1
        fill = torch.linspace(out[-1].item() + step, stop - step, n).to(times.device)
        index[big[k]:] += fill.shape[0]
        out = torch.cat((out, fill, times[big[k]:big[k + 1]].to(times.device)), 0)
    return index, out


class Hypercube:
    """[bot, top]^d x [T0, T]; time-independent domain.  `top_bot` = (bot, top)."""

    def __init__(self, top_bot, dim, T0, T, N_t):
        assert top_bot[1] > top_bot[0], "The hypercube needs to have volume"
        self.bot, self.top = top_bot[0], top_bot[1]
        self.dim, self.T0, self.T, self.N_t = dim, T0, T, N_t
        self.times = _time_grid(T0, T, N_t)

    def _uniform_points(self, n):
        return torch.Tensor(n, 1, self.dim).uniform_(self.bot, self.top).view(n, self.dim)

    def interior(self, N_r):
        return _paths(self.times, self._uniform_points(N_r))

    def boundary(self, N_b):
        x = self._uniform_points(N_b)
        self._uniform_points(N_b)          # the reference draws a second, unused batch here (src/dataset.py:263)
        block = int(N_b / self.dim / 2)    # rows per face; the last face takes the remainder
        cuts = [block * i for i in range(2 * self.dim)] + [N_b]
        for axis in range(self.dim):
            x[cuts[2 * axis]:cuts[2 * axis + 1], axis] = self.top
            x[cuts[2 * axis + 1]:cuts[2 * axis + 2], axis] = self.bot
        return _paths(self.times, x[torch.randperm(N_b)])

    def func_w(self, x):
        """distance to the nearest face: min_i min(|top - x_i|, |bot - x_i|); x is [N, L, 1+d]"""
        xs = x[:, :, 1:]
        to_top = torch.min(torch.abs(self.top - xs), dim=2).values
        to_bot = torch.min(torch.abs(self.bot - xs), dim=2).values
        return torch.minimum(to_top, to_bot)

    def bound_pad(self, x):
        t = torch.cat((torch.tensor(self.T0).view(1).to(x.device), x[0, :, 0]), 0)
        idx, data = fillt(t, self.T, self.T0, self.N_t)
        return None, idx[1:], data

    def V(self):
        return (self.top - self.bot) ** self.dim * (self.T - self.T0)

    # engine hints: the weight w does not depend on time and paths are vertical lines over one shared grid
    time_independent = True


class Comb_loader(Dataset):
    """Groups of equal-length paths: (interior for u, interior for v, boundary).  For a single-tensor domain (cube) the
    v sample is a second, independent interior draw; for list domains it is a copy of the u sample."""

    def __init__(self, N_r, N_b, shape, device):
        self.N_r, self.N_b, self.shape, self.device = N_r, N_b, shape, device
        inner = shape.interior(N_r)
        if isinstance(inner, list):
            self.interioru = [g.requires_grad_(True) for g in inner]
            self.interiorv = [g.clone().detach().requires_grad_(True) for g in self.interioru]
        else:
            self.interioru = inner.requires_grad_(True)
            self.interiorv = shape.interior(N_r).clone().detach().requires_grad_(True)
        edge = shape.boundary(N_b)
        self.boundary = [g.requires_grad_(True) for g in edge] if isinstance(edge, list) else edge.requires_grad_(True)

    def __len__(self):
        return len(self.interioru) if isinstance(self.interioru, list) else 1

    def __getitem__(self, idx):
        if isinstance(self.interioru, list):
            # like the reference, iteration ends at the first IndexError (fewer boundary groups than interior groups
            # silently truncates the epoch)
            group = (self.interioru[idx], self.interiorv[idx], self.boundary[idx])
        else:
            if idx != 0:
                raise IndexError
            group = (self.interioru, self.interiorv, self.boundary)
        return tuple(g.to(self.device) for g in group)


DOMAINS = {'Hypercube': Hypercube}


def resolve_domain(name):
    """The reference does eval(params['domain']) (src/training.py:84); this is a registry lookup instead."""
    if not isinstance(name, str):
        return name
    if name not in DOMAINS:
        raise KeyError('unknown domain %r (registered: %s)' % (name, sorted(DOMAINS)))
    return DOMAINS[name]
