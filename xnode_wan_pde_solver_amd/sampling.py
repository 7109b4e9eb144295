"""Monte-Carlo sampling of space-time paths: domain shapes and the loader, with the reference's call surface and its
exact RNG consumption (so that a seeded run draws the same points; pinned by tests/golden).

Mirrors (file:line into the reference):
    Hypercube            src/dataset.py:232-290
    NSphere_TCone        src/dataset.py:162-229     ball of radius r (1 - t): paths leave the domain, groups by life time
    NSphere_THourglass   src/dataset.py:48-159      ball of radius r (1 - t) then r t: paths may leave and re-enter
    Comb_loader          src/dataset.py:293-322
    fillt                src/dataset.py:13-32
Domain protocol (src/dataset.py:34-45): interior(N_r), boundary(N_b), func_w(x), bound_pad(x), V().
Data layout: [N, L, 1+d], time in channel 0; on the cube every path is one spatial point repeated over the shared,
sorted random time grid.

Sampling stays on the host RNG (torch CPU generator) -- it is O(N d) work per outer iteration and must be
draw-for-draw identical to the reference for "same seeds" parity; the engine uploads only what the kernels need.
"""
import math
import threading
from itertools import groupby

import numpy as np
import torch
from torch.utils.data import Dataset


HIP_HOST_LOCK = threading.Lock()
"""Held by whoever allocates page-locked host memory or captures / launches a HIP graph.  The training loop stages the next
sample on a helper thread while the main thread captures and replays sub-step graphs; nothing else orders a hipHostMalloc of
the one against a capture or launch of the other.  A precaution (the one runtime fault met in round 3 turned out to be graph
DESTRUCTION, engine._KEPT_GRAPHS): the lock is uncontended except during a ring's first use."""


class _PinPool:
    """page-locked staging buffers, a ring per (shape, dtype).  pin_memory() allocates (1 ms for the three tensors of a
    cube sample); copying into a buffer that already exists is 30 us.  The whole ring of a shape is allocated at its first
    use, under HIP_HOST_LOCK.  A slot is reused after RING stagings of its shape (at the headline shape xu, xv and xb share
    one key: 6 stagings per outer iteration, i.e. after 2.7 iterations); what makes that safe is not the count but the
    event `uploaded()` records behind the asynchronous copy out of the slot: `stage()` waits for it before overwriting."""
    RING = 32

    def __init__(self):
        self.bufs, self.pos, self.events, self.mine = {}, {}, {}, set()

    def stage(self, t):
        key = (tuple(t.shape), t.dtype)
        ring = self.bufs.get(key)
        if ring is None:
            with HIP_HOST_LOCK:
                ring = self.bufs[key] = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for _ in range(self.RING)]
                self.mine.update(b.untyped_storage().data_ptr() for b in ring)
        i = self.pos.get(key, 0)
        self.pos[key] = i + 1
        buf = ring[i % self.RING]
        ev = self.events.pop(buf.untyped_storage().data_ptr(), None)
        if ev is not None:
            ev.synchronize()            # (the upload that last read this slot; long done unless the pipeline got deeper)
        buf.copy_(t)
        return buf

    def uploaded(self, buf, stream=None):
        """called right behind an asynchronous host -> device copy out of a staged buffer (solver._up)"""
        self.uploaded_all((buf,), stream)

    def uploaded_all(self, bufs, stream=None):
        """the same for several buffers whose copies were issued on ONE stream: one event behind the last of them guards every
        slot (an event per buffer was 6 x 40 us of an outer iteration: torch looks the current device up for each record)"""
        # (by storage: a rank's share of a staged sample is a slice of the slot)
        ptrs = [b.untyped_storage().data_ptr() for b in bufs if not b.is_cuda and b.untyped_storage().data_ptr() in self.mine]
        if ptrs:
            ev = torch.cuda.Event()
            ev.record(stream) if stream is not None else ev.record()
            for p in ptrs:
                self.events[p] = ev


_PIN_POOL = _PinPool()


class _ArenaPool:
    """page-locked float64 arenas for the samples of the LIST domains (11-20 groups of new shapes every sample: a ring per
    shape would never hit).  A sample's groups are packed into one arena (Comb_loader.pin, on the sampling thread) and travel
    in ONE asynchronous copy instead of ~40 blocking copies out of pageable memory, each of which made the host wait for
    everything queued on the device.  Same protocol as _PinPool: the consumer records an event behind its copy (`uploaded`),
    `take` waits for it before the arena is packed again."""
    RING = 6

    def __init__(self):
        self.bufs, self.events, self.pos = [None] * self.RING, [None] * self.RING, 0

    def take(self, numel):
        i = self.pos % self.RING
        self.pos += 1
        ev, self.events[i] = self.events[i], None
        if ev is not None:
            ev.synchronize()
        if self.bufs[i] is None or self.bufs[i].numel() < numel:
            with HIP_HOST_LOCK:
                self.bufs[i] = torch.empty(max(numel * 5 // 4, 1 << 16), dtype=torch.float64).pin_memory()
        return i, self.bufs[i]

    def uploaded(self, i, stream):
        ev = torch.cuda.Event()
        ev.record(stream)
        self.events[i] = ev


_ARENAS = _ArenaPool()


class _UniformFill:
    """Tensor.uniform_(lo, hi) on torch's default CPU generator for float32 tensors, through the native fill of the C library
    (xw_mt19937_uniform_f32: the generator's state blob is advanced in vectorised loops, 8 x faster than torch's scalar walk --
    the draws of a training iteration are what bounds train() once the GPU side is pipelined).  The FIRST use compares the
    native stream with torch's on this machine, values and final state, across regeneration boundaries; if they differ in any
    bit (another torch build, another state layout), or the library is absent, every fill goes through torch: the numbers a
    seed produces never depend on which way they were drawn.  XW_NATIVE_RNG=0 turns the native fill off."""
    MIN = 2048          # below this torch's own call is as fast as the state round trip

    def __init__(self):
        self.mode = None       # None: not probed yet; False: torch; 0 / 1: native (unfused / fused final multiply-add)

    def _native(self, t, lo, hi, fused):
        st = torch.get_rng_state()
        rc = self.fn(st.data_ptr(), st.numel(), t.data_ptr(), t.numel(), float(lo), float(hi), fused)
        if rc != 0:
            raise RuntimeError('xw_mt19937_uniform_f32 refused the generator state (%d)' % rc)
        torch.set_rng_state(st)
        return t

    def _probe(self):
        import os
        self.mode = False
        if os.environ.get('XW_NATIVE_RNG', '1') != '1':
            return
        try:
            from ._lib import lib
            self.fn = lib.xw_mt19937_uniform_f32
        except Exception:
            return
        keep = torch.get_rng_state()
        try:
            for fused in (1, 0):
                ok = True
                torch.default_generator.manual_seed(20240229)      # (the CPU generator only: torch.manual_seed would reseed the GPUs')
                for n, lo, hi in ((5, -1.0, 1.0), (700, 0.0, 1.0), (623, -3.0, 0.5), (625, -1.0, 2.0), (40000, -1.5, 1.5), (3, 0.0, 1.0)):
                    s0 = torch.get_rng_state()
                    a = torch.empty(n).uniform_(lo, hi)
                    s1 = torch.get_rng_state()
                    torch.set_rng_state(s0)
                    b = self._native(torch.empty(n), lo, hi, fused)
                    ok = ok and torch.equal(a, b) and torch.equal(s1, torch.get_rng_state())
                if ok:
                    self.mode = fused
                    break
        except Exception:
            self.mode = False
        finally:
            torch.set_rng_state(keep)
        if self.mode is False:
            import warnings
            # (not silent: the draws of an outer iteration then take 8 x longer and bound train() again -- same numbers either way)
            warnings.warn("the native uniform fill does not reproduce this torch build's CPU generator stream (torch %s): the samplers draw "
                          'through torch.Tensor.uniform_ (same numbers, ~8 x slower draws)' % torch.__version__, RuntimeWarning)

    def __call__(self, t, lo, hi):
        if self.mode is None:
            self._probe()
        if (self.mode is False or t.numel() < self.MIN or t.dtype != torch.float32 or t.device.type != 'cpu'
                or not t.is_contiguous()):
            return t.uniform_(lo, hi)
        return self._native(t, lo, hi, self.mode)


_uniform_fill = _UniformFill()


class _LegacyNormal:
    """np.random.normal(size=shape) on numpy's GLOBAL legacy generator through the native fill of the C library
    (xw_mt19937_legacy_normal_f64: numpy walks legacy_gauss value by value, 16-18 ns each, and a ball-domain sample draws 0.1-0.3 M
    of them -- the draws were what bounded train() on those domains).  The FIRST use compares the native stream with numpy's,
    values and the state left behind (key, position, cached second value), over sizes that cross state blocks and leave /
    consume a cached value; if anything differs, or the library is absent, every draw goes through numpy: the numbers a seed
    produces never depend on which way they were drawn.  XW_NATIVE_RNG=0 turns the native fill off."""
    MIN = 4096          # below this numpy's own call is as fast as the state round trip (get_state + set_state: 80 us)

    def __init__(self):
        import threading
        self.ok = None
        self._local = threading.local()

    def _fill(self, st, shape):
        import ctypes
        out = np.empty(shape)
        p, h, c = ctypes.c_int(st[2]), ctypes.c_int(st[3]), ctypes.c_double(st[4])
        rc = self.fn(st[1].ctypes.data, ctypes.byref(p), ctypes.byref(h), ctypes.byref(c), out.ctypes.data, out.size)
        if rc != 0:
            raise RuntimeError("xw_mt19937_legacy_normal_f64 refused numpy's generator state (%d)" % rc)
        st[2:] = p.value, h.value, c.value
        return out

    def _native(self, shape):
        st = list(np.random.get_state())
        if st[0] != 'MT19937' or st[1].dtype != np.uint32 or not st[1].flags.c_contiguous:
            return np.random.normal(size=shape)
        out = self._fill(st, shape)
        np.random.set_state(tuple(st))
        return out

    def hold(self):
        """context: numpy's global state is taken ONCE, every draw inside goes through the native fill on that copy whatever its
        size, and the state goes back at the end (a boundary sample is 20 draws; the round trip would be a third of their time).
        Nothing else may draw from numpy's global generator inside."""
        from contextlib import contextmanager

        @contextmanager
        def held():
            if self.ok is None:
                self._probe()
            st = list(np.random.get_state()) if self.ok else None
            if st is None or st[0] != 'MT19937' or st[1].dtype != np.uint32 or not st[1].flags.c_contiguous:
                yield
                return
            self._local.state = st
            try:
                yield
            finally:
                self._local.state = None
                np.random.set_state(tuple(st))
        return held()

    def _probe(self):
        import os
        self.ok = False
        if os.environ.get('XW_NATIVE_RNG', '1') != '1':
            return
        try:
            from ._lib import lib
            self.fn = lib.xw_mt19937_legacy_normal_f64
        except Exception:
            return
        keep = np.random.get_state()
        try:
            same = True
            for seed in (20240229, 7):
                sizes = (5, 1, 700, 155, 156, 157, 312, 313, (10, 4099), 3, 2, 40001, 1)
                np.random.seed(seed)
                a = [np.random.normal(size=n) for n in sizes]
                sa = np.random.get_state()
                np.random.seed(seed)
                b = [self._native(n) for n in sizes]
                sb = np.random.get_state()
                same = (same and all(x.shape == y.shape and np.array_equal(x, y) for x, y in zip(a, b))
                        and np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:])
            self.ok = bool(same)
        except Exception:
            self.ok = False
        finally:
            np.random.set_state(keep)
        if not self.ok:
            import warnings
            warnings.warn("the native normal fill does not reproduce this numpy build's legacy generator stream (numpy %s): the ball "
                          'domains draw through np.random.normal (same numbers, ~2 x slower draws)' % np.__version__, RuntimeWarning)

    def __call__(self, shape):
        st = getattr(self._local, 'state', None)
        if st is not None:
            return self._fill(st, shape)
        if self.ok is None:
            self._probe()
        if not self.ok or int(np.prod(shape)) < self.MIN:
            return np.random.normal(size=shape)
        return self._native(shape)


_legacy_normal = _LegacyNormal()
_FACE_TABLES = {}


def _time_grid(T0, T, N_t):
    grid, _ = torch.sort(torch.Tensor(N_t).uniform_(T0, T), 0)
    grid[0], grid[-1] = T0, T
    return grid


def _paths(times, x):
    """[n, d] spatial points -> [n, L, 1+d] vertical paths over the time grid (same device / dtype as x)."""
    n, d = x.shape
    L = times.shape[0]
    t = times.to(device=x.device, dtype=x.dtype)
    return torch.cat((t.view(1, L, 1).expand(n, L, 1), x.view(n, 1, d).expand(n, L, d)), 2)


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
        return _uniform_fill(torch.Tensor(n, 1, self.dim), self.bot, self.top).view(n, self.dim)

    def interior_x(self, N_r):
        """the N_r spatial points of an interior sample, [N_r, d] (compact form of interior())"""
        return self._uniform_points(N_r)

    def interior(self, N_r):
        return _paths(self.times, self.interior_x(N_r))

    def boundary(self, N_b):
        return _paths(self.times, self.boundary_x(N_b))

    def boundary_x(self, N_b):
        x = self._uniform_points(N_b)
        self._uniform_points(N_b)          # the reference draws a second, unused batch here (src/dataset.py:263)
        rows, axis, val = self._faces(N_b)
        x[rows, axis] = val                # row k sits on face k // block: top of axis 0, bottom of axis 0, top of axis 1, ...
        return x[torch.randperm(N_b)]

    def _faces(self, N_b):
        """(row, axis, value) of the pinned coordinate of every boundary row before the shuffle (src/dataset.py:265-272): blocks of
        int(N_b / d / 2) rows per face, the last face takes the remainder; cached per N_b (one scatter instead of 2 d slices)"""
        key = (N_b, self.dim, float(self.top), float(self.bot))
        tab = _FACE_TABLES.get(key)              # (module-level: the training loop builds a new domain object per sample)
        if tab is None:
            block = int(N_b / self.dim / 2)
            cuts = [block * i for i in range(2 * self.dim)] + [N_b]
            rows = torch.arange(N_b)
            face = torch.zeros(N_b, dtype=torch.long)
            for f in range(2 * self.dim):
                face[cuts[f]:cuts[f + 1]] = f
            val = torch.where(face % 2 == 0, torch.tensor(float(self.top)), torch.tensor(float(self.bot)))
            if len(_FACE_TABLES) > 64:
                _FACE_TABLES.clear()
            tab = _FACE_TABLES[key] = (rows, face // 2, val)
        return tab

    def device_sample(self, N_r, N_b, device):
        """(x_u, x_v, x_b) drawn with the DEVICE generator: same distribution as interior/interior/boundary, no seed
        parity with the reference's host draws (NODE_WAN_solver.device_sampling)"""
        span = self.top - self.bot
        xu = torch.rand(N_r, self.dim, device=device) * span + self.bot
        xv = torch.rand(N_r, self.dim, device=device) * span + self.bot
        xb = torch.rand(N_b, self.dim, device=device) * span + self.bot
        block = int(N_b / self.dim / 2)
        cuts = [block * i for i in range(2 * self.dim)] + [N_b]
        for axis in range(self.dim):
            xb[cuts[2 * axis]:cuts[2 * axis + 1], axis] = self.top
            xb[cuts[2 * axis + 1]:cuts[2 * axis + 2], axis] = self.bot
        return xu, xv, xb[torch.randperm(N_b, device=device)]

    def func_w(self, x):
        """distance to the nearest face: min_i min(|top - x_i|, |bot - x_i|); x is [N, L, 1+d]"""
        xs = x[:, :, 1:]
        to_top = torch.min(torch.abs(self.top - xs), dim=2).values
        to_bot = torch.min(torch.abs(self.bot - xs), dim=2).values
        return torch.minimum(to_top, to_bot)

    def func_w_grad(self, x):
        """(w, dw/d[t, x]) in closed form -- what torch.autograd.grad(func_w(x).sum(), x) returns, entry for entry, ties
        included: min over a dimension sends the gradient to the index it reports, torch.minimum splits it half / half
        on equal operands, |.| has slope sign(.) (0 at 0).  The engine calls this instead of autograd once per sample
        (0.5 of 4.5 ms per outer iteration at the headline size)."""
        xs = x[:, :, 1:]
        dt_, db_ = self.top - xs, self.bot - xs
        to_top, i_top = torch.min(torch.abs(dt_), dim=2)
        to_bot, i_bot = torch.min(torch.abs(db_), dim=2)
        w = torch.minimum(to_top, to_bot)
        c_top = (to_top < to_bot).to(x.dtype) + 0.5 * (to_top == to_bot).to(x.dtype)
        c_bot = (to_bot < to_top).to(x.dtype) + 0.5 * (to_top == to_bot).to(x.dtype)
        g = torch.zeros_like(x)
        gx = g[:, :, 1:]
        gx.scatter_add_(2, i_top.unsqueeze(2), (-torch.sign(torch.gather(dt_, 2, i_top.unsqueeze(2))) * c_top.unsqueeze(2)))
        gx.scatter_add_(2, i_bot.unsqueeze(2), (-torch.sign(torch.gather(db_, 2, i_bot.unsqueeze(2))) * c_bot.unsqueeze(2)))
        return w, g

    def bound_pad(self, x):
        t = torch.cat((torch.tensor(self.T0).view(1).to(x.device), x[0, :, 0]), 0)
        idx, data = fillt(t, self.T, self.T0, self.N_t)
        return None, idx[1:], data

    def V(self):
        return (self.top - self.bot) ** self.dim * (self.T - self.T0)

    # engine hints: the weight w does not depend on time and paths are vertical lines over one shared grid
    time_independent = True


def _ball_volume_factor(dim, r):
    from scipy.special import gamma
    return math.pi ** (dim / 2) / gamma(dim / 2 + 1) * r ** dim


class _NSphereBase:
    """Shared pieces of the two time-varying ball domains.  Spatial points come from numpy's GLOBAL generator (like the
    reference: seed with np.random.seed), the time grid from torch's.  Paths are float64; a group is a tensor
    [n, L_k, 1+d] of paths with the same number of samples; boundary groups hold one time each, [n_t, 1, 1+d]."""
    time_independent = False

    def __init__(self, r, dim, T0, T, N_t):
        self.r, self.dim, self.T0, self.T, self.N_t = r, dim, T0, T, N_t
        self.times = _time_grid(T0, T, N_t)

    def surf(self, N):
        """N points uniform on the sphere of radius r, as a [dim, N] array"""
        z = _legacy_normal((self.dim, N))
        return self.r * z / np.sqrt((z ** 2).sum(axis=0))

    def _ball(self, N):
        pts = self.surf(N)
        pts *= np.random.rand(N) ** (1 / self.dim)
        return pts

    def _shell_groups(self, N_b, radius_factor):
        """one boundary group per sample time: int(N_b * factor(t)^dim) points on the sphere of radius factor(t)
        (the reference scales the unit-r surface by factor(t) only, src/dataset.py:103,196)"""
        out = []
        with _legacy_normal.hold():
            for t in self.times.numpy():
                fac = radius_factor(t)
                n = int(N_b * fac ** self.dim)
                pts = torch.from_numpy(self.surf(n) * fac).t().unsqueeze(1)
                if n != 0:
                    out.append(torch.cat((t * torch.ones(n, 1, 1), pts), 2).requires_grad_(True))
        return out


    def skip_boundary(self, N_b):
        """the draws of boundary(N_b) -- numpy's stream ends where it would -- without the points: the post-step diagnostic
        sample (src/training.py:166-167) is drawn in full by the reference but only its interior is ever read"""
        with _legacy_normal.hold():
            for t in self.times.numpy():
                _legacy_normal((self.dim, int(N_b * self._radius_factor(t) ** self.dim)))


class NSphere_TCone(_NSphereBase):
    """{ |x| < r (1 - t) }: the ball shrinks linearly to a point at t = 1."""

    def interior(self, N_r):
        pts = self._ball(N_r)
        tcol = self.times.repeat(N_r, 1).unsqueeze(2)
        groups, k = [], self.N_t
        # (the reference recomputes the norms of the points still unassigned and np.delete()s the chosen ones at every
        #  sample time, src/dataset.py:170-185; the norms never change and deletion keeps the order: one norm per point
        #  and a mask of the unassigned ones give the same groups, bit for bit, in a third of the time)
        nrm = np.sqrt(np.sum(pts ** 2, 0))
        left = np.ones(N_r, dtype=bool)
        for t in self.times.numpy()[::-1]:
            # walking back from T: a point first found inside at this time lives for the k samples t_0 .. t_{k-1}
            alive = left & (nrm < self.r * (1 - t))
            chosen = torch.from_numpy(pts[:, alive]).t().unsqueeze(1).repeat(1, k, 1)
            left &= ~alive
            if chosen.shape[0] != 0:
                groups.append(torch.cat((tcol[:chosen.shape[0], :k], chosen), 2).requires_grad_(True))
            k -= 1
        return groups[::-1]

    def _radius_factor(self, t):
        return 1 - t

    def boundary(self, N_b):
        return self._shell_groups(N_b, self._radius_factor)

    def func_w(self, x):
        return self.r * (1 - x[:, :, 0]) - torch.sqrt(torch.sum(x[:, :, 1:] ** 2, 2))

    def bound_pad(self, x):
        t = torch.cat((torch.tensor(self.T0).view(1), x[0, :, 0]), 0)
        idx, data = fillt(t, self.T, self.T0, self.N_t)
        return None, idx[1:], data

    def V(self):
        d1 = self.dim + 1
        return _ball_volume_factor(self.dim, self.r) * ((1 - self.T0) ** d1 / d1 - (1 - self.T) ** d1 / d1)


class NSphere_THourglass(_NSphereBase):
    """{ |x| < r ((T - T0) - t) } for t <= (T - T0)/2, { |x| < r t } after: paths can leave the domain and re-enter; a
    re-entering piece is prepended its entry point on the moving boundary (time |x| / r)."""

    def _half(self):
        return (self.T - self.T0) / 2

    def interior(self, N_r):
        """Same groups, same order and the same bits as the reference's per-path loop (src/dataset.py:62-104), without the
        loop: a path is inside for its first n_first samples, outside for a contiguous middle stretch, inside again for
        its last n_late samples.  Groups = pieces bucketed by length (paths in sample order within a bucket), first pieces
        before re-entry pieces of the same length."""
        pts = self._ball(N_r)
        L = self.N_t
        # The reference materialises every path as [N, L, 1 + d] float64 and takes norms, bounds and masks on that tensor
        # (src/dataset.py:88-96).  A path is one point repeated over the time grid: ONE norm per path and one bound per
        # time index give the same mask bit for bit (same reduction over the same d contiguous numbers; the bound is
        # float32 arithmetic on the float32 grid, then widened, as there), and the groups are assembled from [N, d] and [L]
        # directly -- 15 MB of traffic per sample instead of 100.
        X = torch.from_numpy(pts).t().contiguous()                                        # [N, d] float64
        t32 = self.times
        early = torch.le(t32, self._half())
        bound = torch.where(early, self.r * ((self.T - self.T0) - t32).double(), self.r * t32.double())   # [L]  (float32 difference, widened, THEN times r in float64: src/dataset.py:89-90 -- `.double()` binds before `*`)
        nrm = torch.sqrt(torch.sum(X ** 2, 1))
        inside = nrm.view(N_r, 1) < bound.view(1, L)
        t64 = t32.double()
        out = ~inside
        any_out = out.any(1)
        idx = torch.arange(L).expand(N_r, L)
        first_out = torch.where(out, idx, torch.full_like(idx, L)).min(1).values        # = L when the path never leaves
        last_out = torch.where(out, idx, torch.full_like(idx, -1)).max(1).values
        n_first = torch.where(any_out, first_out, torch.full_like(first_out, L))
        n_late = torch.where(any_out, L - last_out - 1, torch.zeros_like(last_out))
        if bool((out.sum(1) != (L - n_first - n_late)).any()):
            raise RuntimeError('a sampled path leaves the hourglass more than once')        # (the reference's split would raise)

        def piece(ks, lo, hi):          # samples lo .. hi - 1 of the paths ks, [len(ks), hi - lo, 1 + d]
            k, n = ks.shape[0], hi - lo
            return torch.cat((t64[lo:hi].view(1, n, 1).expand(k, n, 1), X[ks].view(k, 1, -1).expand(k, n, X.shape[1])), 2)
        # entry point of the re-entering pieces on the moving boundary: time |x| / r
        t_in = torch.sqrt(torch.sum(X[any_out] ** 2, 1)) / self.r
        entry = torch.cat((t_in.view(-1, 1, 1), X[any_out].unsqueeze(1)), 2)               # [n_out, 1, 1 + d]
        late_rows = any_out.nonzero().view(-1)
        groups_first, groups_late = [], []
        for ell in sorted(set(n_first.tolist())):
            ks = (n_first == ell).nonzero().view(-1)
            groups_first.append(piece(ks, 0, ell))
        nl = n_late[late_rows]
        for ell in sorted(set(nl.tolist())):
            sel = (nl == ell).nonzero().view(-1)
            ks = late_rows[sel]
            groups_late.append(torch.cat((entry[sel], piece(ks, L - ell, L)), 1))
        return sorted([*groups_first, *groups_late], key=lambda g: g.shape[1])

    def _radius_factor(self, t):
        return (self.T - self.T0 - t) if t < self._half() else t

    def boundary(self, N_b):
        return self._shell_groups(N_b, self._radius_factor)

    def func_w(self, x):
        t = x[:, :, 0]
        dist = torch.sqrt(torch.sum(x[:, :, 1:] ** 2, 2))
        early = torch.le(t, self._half())
        # (a select, not two masked assignments: same value and same gradient per point, and nothing waits for the device -- a
        #  boolean-mask index is a nonzero() whose size the host has to read back)
        return torch.where(early, self.r * ((self.T - self.T0) - t) - dist, self.r * t - dist)

    def bound_pad(self, x):
        """Per-path densified grids for the evaluation of paths that start neither at T0 nor on the boundary
        (reference src/dataset.py:127-152).  A path whose first sample lies in the widening half is integrated from its
        entry point on the moving boundary (time |x| / r, or T0 if it never left); paths are bucketed by the length of
        their filled grid and -- like the reference -- every bucket uses the grid and the index map of its FIRST path.
        Returns (path indices per bucket, index map per bucket, grid per bucket)."""
        t_first = x[0, 0, 0]
        on_bdry = None
        if t_first < self._half():
            t_in = self.T0 * torch.ones_like(x[:, 0, 0])
        else:
            rad = torch.sqrt(torch.sum(x[:, 0, 1:] ** 2, dim=-1))
            never_left = torch.le(rad, self.r * self._half())
            t_in = torch.zeros_like(x[:, 0, 0])
            t_in[never_left] = self.T0
            t_in[~never_left] = rad[~never_left] / self.r
            on_bdry = torch.le(self.func_w(x[:, 0].unsqueeze(1)), 1e-5).squeeze().int()
        t_all = torch.cat((t_in.unsqueeze(1), x[:, :, 0]), dim=1)
        n = t_all.shape[0]
        grids = [fillt(t_all[k], self.T, self.T0, self.N_t)[1] for k in range(n)]
        if on_bdry is None:
            maps = [fillt(t_all[k], self.T, self.T0, self.N_t)[0] for k in range(n)]
        else:   # a first sample that sits on the boundary is its own entry point: drop the prepended time for the map
            maps = [fillt(t_all[k][int(on_bdry[k]):], self.T, self.T0, self.N_t)[0][1 - int(on_bdry[k]):] for k in range(n)]
        order = sorted(range(n), key=lambda k: grids[k].shape[0])
        path_i, idx, data = [], [], []
        for _, bucket in groupby(order, key=lambda k: grids[k].shape[0]):
            ks = list(bucket)
            path_i.append(torch.tensor(ks))
            idx.append(maps[ks[0]])
            data.append(grids[ks[0]])
        return path_i, idx, data

    def V(self):
        d1 = self.dim + 1
        return _ball_volume_factor(self.dim, self.r) * 2 * ((1 - self.T0) ** d1 / d1 - (1 - self._half()) ** d1 / d1)


class Comb_loader(Dataset):
    """Groups of equal-length paths: (interior for u, interior for v, boundary).  For a single-tensor domain (cube) the
    v sample is a second, independent interior draw; for list domains it is a copy of the u sample.

    For domains that offer compact draws (`interior_x` / `boundary_x`: vertical paths over one shared grid) only the
    [N, d] points are drawn here -- same RNG consumption -- and the [N, L, 1+d] tensors `interioru`, `interiorv`,
    `boundary` are materialised on first access; the engine works from the compact form (`compact()`)."""

    def __init__(self, N_r, N_b, shape, device, interior_only=False):
        """interior_only: the sample is drawn in full -- every generator ends where it would -- but a list domain that can
        consume its boundary draws without building the points (skip_boundary) does so, and `boundary` is empty: the post-step
        diagnostic sample, of which only the interior is read."""
        self.N_r, self.N_b, self.shape, self.device = N_r, N_b, shape, device
        self._lazy, self._cache = None, {}
        if hasattr(shape, 'interior_x') and hasattr(shape, 'boundary_x'):
            self._lazy = {'interioru': shape.interior_x(N_r), 'interiorv': shape.interior_x(N_r), 'boundary': shape.boundary_x(N_b)}
            return
        inner = shape.interior(N_r)
        if isinstance(inner, list):
            self._cache['interioru'] = [g.requires_grad_(True) for g in inner]       # (interiorv: a copy, made when first asked for)
            if interior_only and hasattr(shape, 'skip_boundary'):
                shape.skip_boundary(N_b)
                self._cache['boundary'] = []
                return
        else:
            self._cache['interioru'] = inner.requires_grad_(True)
            self._cache['interiorv'] = shape.interior(N_r).clone().detach().requires_grad_(True)
        edge = shape.boundary(N_b)
        self._cache['boundary'] = [g.requires_grad_(True) for g in edge] if isinstance(edge, list) else edge.requires_grad_(True)

    def _get(self, name):
        if name not in self._cache:
            if self._lazy is None:        # list domains: the v sample is a copy of the u sample (14 MB at config-5 size; the engine reads the u sample for both)
                self._cache[name] = [g.clone().detach().requires_grad_(True) for g in self._cache['interioru']]
            else:
                self._cache[name] = _paths(self.shape.times, self._lazy[name]).requires_grad_(True)
        return self._cache[name]

    interioru = property(lambda self: self._get('interioru'))
    interiorv = property(lambda self: self._get('interiorv'))
    boundary = property(lambda self: self._get('boundary'))

    def compact(self):
        """(times[L], x_u[N,d], x_v[N,d], x_b[N_b,d]) or None when the domain has no compact form"""
        if self._lazy is None:
            return None
        return getattr(self, '_times_pinned', self.shape.times), self._lazy['interioru'], self._lazy['interiorv'], self._lazy['boundary']

    def pin(self):
        """page-lock the compact sample (same values): its upload can then be asynchronous.  A copy from pageable memory makes
        the host wait for everything already queued on the device -- in the training loop that is a whole outer iteration.
        List domains: the groups are packed into one page-locked arena (pack)."""
        if self._lazy is not None and torch.cuda.is_available():
            self._lazy = {k: _PIN_POOL.stage(v) for k, v in self._lazy.items()}
            self._times_pinned = _PIN_POOL.stage(self.shape.times)
        elif self._lazy is None and torch.cuda.is_available() and isinstance(self._cache.get('interioru'), list):
            self.pack(pinned=True)
        return self

    def pack(self, pinned=False, into=None):
        """List domains: all interior groups (u; the v sample is a copy of it) and boundary groups of the sample in ONE flat
        float64 buffer, with what the engine otherwise reads back from the device group by group, taken from the host copies
        here: the first time of every group, whether its paths share one time column, whether a boundary group sits on its
        interior group's grid.  Returns (buffer, layout) and keeps them; `device_groups` uploads the buffer in one copy."""
        if getattr(self, '_packed', None) is not None:
            return self._packed
        gu, gb = self._cache['interioru'], self._cache['boundary']
        n = min(len(gu), len(gb))                             # (iteration stops at the shorter list: __getitem__)
        tensors = [g.detach() for g in gu] + [g.detach() for g in gb]
        if any(t.dtype != torch.float64 for t in tensors):
            self._packed = False
            return False
        total = sum(t.numel() for t in tensors)
        if into is not None:              # (a caller's buffer: the sampling process packs into shared memory, sampler_proc.py)
            slot, buf = None, into
        else:
            slot, buf = _ARENAS.take(total) if pinned else (None, torch.empty(total, dtype=torch.float64))
        offs, o = [], 0
        for t in tensors:
            buf[o:o + t.numel()].view(t.shape).copy_(t)
            offs.append((o, tuple(t.shape)))
            o += t.numel()
        hints = []
        for k in range(len(gu)):
            x = tensors[k]
            h = dict(t0=float(x[0, 0, 0]), shared_times=bool(torch.all(x[:, :, 0] == x[:1, :, 0])))
            if k < n:
                b = tensors[len(gu) + k]
                h.update(tb0=float(b[0, 0, 0]), same_grid=b.shape[1] == x.shape[1] and bool(torch.equal(b[0, :, 0], x[0, :, 0])))
            hints.append(h)
        self._packed = (slot, buf, total, offs, hints, n, len(gu))
        return self._packed

    def _device_views(self, device):
        if self._lazy is not None or not isinstance(self._cache.get('interioru'), list):
            return None
        packed = self.pack()
        if not packed:
            return None
        if getattr(self, '_views', None) is None:
            slot, buf, total, offs = packed[:4]
            dev = buf[:total].to(device, non_blocking=True)
            if slot is not None:
                _ARENAS.uploaded(slot, torch.cuda.current_stream(device))
            self._views = [dev[o:o + math.prod(shape)].view(shape) for o, shape in offs]
        return self._views

    def device_groups(self, device):
        """[(X, XV, BX)] on the device -- views of one uploaded buffer (XV is X: the v sample of a list domain is a copy of the u
        sample) -- and the per-group hints of pack(); None when the sample is not a packable list sample"""
        views = self._device_views(device)
        if views is None:
            return None
        hints, n, nu = self._packed[4:]
        return [(views[k], views[k], views[nu + k]) for k in range(n)], hints[:n]

    def device_interior(self, device):
        """ALL interior groups on the device (the diagnostic integrates over every one of them, also where the boundary list
        is shorter) with their hints; None as above"""
        views = self._device_views(device)
        if views is None:
            return None
        hints, n, nu = self._packed[4:]
        return views[:nu], hints

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


class DeviceCubeLoader:
    """Comb_loader's attributes for a sample drawn on the device (Hypercube.device_sample); no host tensors at all."""

    def __init__(self, N_r, N_b, shape, device):
        self.N_r, self.N_b, self.shape, self.device = N_r, N_b, shape, device
        self._x = shape.device_sample(N_r, N_b, device)
        self._cache = {}

    def _get(self, i, name):
        if name not in self._cache:
            self._cache[name] = _paths(self.shape.times, self._x[i]).requires_grad_(True)
        return self._cache[name]

    interioru = property(lambda self: self._get(0, 'interioru'))
    interiorv = property(lambda self: self._get(1, 'interiorv'))
    boundary = property(lambda self: self._get(2, 'boundary'))

    def compact(self):
        return (getattr(self, '_times_pinned', self.shape.times),) + tuple(self._x)

    def pin(self):
        if torch.cuda.is_available():
            self._times_pinned = _PIN_POOL.stage(self.shape.times)
            self._x = tuple(x if x.is_cuda else _PIN_POOL.stage(x) for x in self._x)
        return self

    def __len__(self):
        return 1

    def __getitem__(self, idx):
        if idx != 0:
            raise IndexError
        return self.interioru, self.interiorv, self.boundary


class RankCubeLoader:
    """This rank's share of a cube sample, drawn WITHOUT generating the global sample (multi-GPU training).

    Comb_loader on R ranks draws the whole global sample on every rank from the shared seed and keeps a slice: exact seed
    parity with the single-GPU run, but a host cost that grows with the global batch on every rank (65,536 x 100 x 3
    uniforms per resample at BASELINE configs[3]).  Here every rank draws ONE integer from the shared stream (so the
    streams of all ranks stay in step: the time grids of later iterations, which must be common, keep coming out
    identical) and seeds a private generator with (that integer, rank) for its own N/R interior, N/R test-function and
    N_b/R boundary points.  Consequence, documented in DESIGN section 6: same distribution, NOT the same numbers as the
    unsharded draw of that seed -- a run with R ranks is reproducible for that R, not bit-comparable across R."""

    def __init__(self, N_r, N_b, shape, device, rank, size):
        if not (hasattr(shape, 'interior_x') and hasattr(shape, 'boundary_x')):
            raise ValueError('rank-local sampling needs a domain with compact draws (Hypercube)')
        self.N_r, self.N_b, self.shape, self.device, self.rank, self.size = N_r, N_b, shape, device, rank, size
        base = int(torch.randint(0, 2 ** 62, (1,)).item())
        gen = torch.Generator().manual_seed((base + 0x9E3779B97F4A7C15 * (rank + 1)) % (2 ** 63 - 1))
        lo, hi = _bounds(N_r, rank, size)
        blo, bhi = _bounds(N_b, rank, size)
        self.n_local, self.nb_local = hi - lo, bhi - blo
        if self.n_local == 0 or self.nb_local == 0:
            raise RuntimeError('sample of %d/%d paths is too small for %d ranks' % (N_r, N_b, size))
        d, span = shape.dim, shape.top - shape.bot
        draw = lambda n: torch.rand(n, d, generator=gen) * span + shape.bot   # noqa: E731
        xu, xv, xb = draw(self.n_local), draw(self.n_local), draw(self.nb_local)
        # faces: the global sample pins int(N_b / d / 2) points to each of the 2 d faces (the last face takes the
        # remainder) and shuffles; a shard of a shuffled sample has those proportions in expectation: draw the face per point
        block = int(N_b / d / 2)
        cuts = torch.tensor([block * i for i in range(2 * d)] + [N_b])
        face = torch.bucketize(torch.randint(0, N_b, (self.nb_local,), generator=gen), cuts[1:], right=True).clamp_(max=2 * d - 1)
        rows = torch.arange(self.nb_local)
        xb[rows, face // 2] = torch.where(face % 2 == 0, torch.tensor(float(shape.top)), torch.tensor(float(shape.bot)))
        self._x = (xu, xv, xb)
        self._cache = {}

    def _get(self, i, name):
        if name not in self._cache:
            self._cache[name] = _paths(self.shape.times, self._x[i]).requires_grad_(True)
        return self._cache[name]

    interioru = property(lambda self: self._get(0, 'interioru'))
    interiorv = property(lambda self: self._get(1, 'interiorv'))
    boundary = property(lambda self: self._get(2, 'boundary'))

    def compact(self):
        return (getattr(self, '_times_pinned', self.shape.times),) + tuple(self._x)

    def pin(self):
        if torch.cuda.is_available():
            self._times_pinned = _PIN_POOL.stage(self.shape.times)
            self._x = tuple(x if x.is_cuda else _PIN_POOL.stage(x) for x in self._x)
        return self

    def __len__(self):
        return 1

    def __getitem__(self, idx):
        if idx != 0:
            raise IndexError
        return tuple(t.to(self.device) for t in (self.interioru, self.interiorv, self.boundary))


def _bounds(n, rank, size):
    """contiguous, balanced split of n items: rank r owns [lo, hi)   (same rule as dist.World.bounds)"""
    base, rem = divmod(n, size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


DOMAINS = {'Hypercube': Hypercube, 'NSphere_TCone': NSphere_TCone, 'NSphere_THourglass': NSphere_THourglass}


def resolve_domain(name):
    """The reference does eval(params['domain']) (src/training.py:84); this is a registry lookup instead."""
    if not isinstance(name, str):
        return name
    if name not in DOMAINS:
        raise KeyError('unknown domain %r (registered: %s)' % (name, sorted(DOMAINS)))
    return DOMAINS[name]
