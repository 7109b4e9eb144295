"""The fused XNODE-WAN training step: generator and discriminator sub-steps as explicit sequences of HIP kernels.

Replaces the bodies of the two sub-step loops of NODE_WAN_solver.train (src/training.py:125-138,151-162 of the
reference): forward of both nets, func_eval, loss.u / loss.v with their three autograd backward passes, and
optimizer.step().  No autograd tape is built; every gradient is an explicit kernel (xw_ode_bwd, xw_disc_bwd).

Semantics are the reference's ON A GPU, stated once here and implemented literally below:
  Q1  the helper backwards of loss.I also deposit d(sum u)/dtheta resp. d(sum phi)/dphi in the parameter gradients
      -> `pollution = 1` in the cotangent kernels (set Engine.pollution = 0.0 for the textbook gradient);
  Q2  nabla u, nabla phi enter I as constants; the u-factor of the d(phi)/dt term carries no gradient;
  Q3  nabla_x u is the time-summed input gradient G_n = d(sum_l u[n,l])/dx_n (incl. the path through h(x)), seen at
      time index 0 only -> the a_ij contraction is per path (s3x);
  Q4  v is evaluated on its own interior sample XV;
  Q5  no stale .grad carry-over between sub-steps (that is a CPU-only artefact of the reference).

Data layout in HBM (per group of N equal-length paths, L sample times, d dimensions):
  xT, xvT, xbT  float32 [d, N]   transposed coordinates of the u-, v- and boundary samples (the [N, L, d+1] path tensor
                                 of the reference is never materialised on the device: on vertical paths it is x (x) t)
  t             float32 [L]      shared time grid
  u, v, vt, f, ubar, vbar ...    float64 [L, N]   time-major point arrays (coalesced for one-lane-per-path kernels)
  Y             float64 [L, H, N]  hidden-state checkpoints of the stepper (written by the forward, read by the sweeps)
  slabs         float64 [n_slab, P]  per-wave partial parameter gradients, summed inside the Adam kernel
"""
import torch

from . import kernels as KN
from ._lib import XnwanError

F32, F64 = torch.float32, torch.float64


def _to_LN(x, dev):
    """[N, L] (any float dtype, any device) -> contiguous float64 [L, N] on dev"""
    return x.detach().to(dev).to(F64).t().contiguous()


class Structure:
    """What the PDE coefficient callables look like, found by probing them once on a few random points."""

    def __init__(self, funcs, d, lo=-1.0, hi=1.0):
        g = torch.Generator().manual_seed(20221111)
        Xp = torch.rand(6, 3, d + 1, generator=g) * (hi - lo) + lo
        Xp[:, :, 0] = torch.rand(6, 3, generator=g)
        ident = True
        for i in range(d):
            for j in range(d):
                a = funcs['a'](Xp, i, j)
                if not bool(torch.all(a == (1.0 if i == j else 0.0))):
                    ident = False
                    break
            if not ident:
                break
        self.a_identity = ident
        self.b_zero = all(bool(torch.all(funcs['b'](Xp, i) == 0)) for i in range(d))
        u1 = torch.randn(6, 3, 1, generator=g, dtype=F64)
        u2 = torch.randn(6, 3, 1, generator=g, dtype=F64)
        c1, c2 = funcs['c'](Xp, u1), funcs['c'](Xp.flip(0), u2)
        k = (c1 / u1).reshape(-1)
        self.c_kappa = None
        if torch.is_tensor(c1) and c1.shape == u1.shape and bool(torch.all(k == k[0])) and bool(torch.all(c2 == k[0] * u2)):
            self.c_kappa = float(k[0])

    def describe(self):
        return 'a=%s b=%s c=%s' % ('identity' if self.a_identity else 'general', 'zero' if self.b_zero else 'general',
                                   ('%g*u' % self.c_kappa) if self.c_kappa is not None else 'general')


class Group:
    """Device-resident data of one group of equal-length paths + every buffer its sub-steps need (allocated once)."""
    pass


class Engine:
    def __init__(self, config, setup, u_mod, v_mod, funcs, device, world=None, structure=None):
        self.config, self.setup, self.u, self.v, self.funcs, self.dev, self.world = config, setup, u_mod, v_mod, funcs, device, world
        self.d = setup['dim']
        self.H, self.K, self.m = config['u_hidden_dim'], config['u_hidden_hidden_dim'], config['u_layers']
        self.W, self.q = config['v_hidden_dim'], config['v_layers']
        self.method = KN.method_id(config['solver'])
        self.alpha = float(config['alpha'])
        self.pollution = 1.0
        sp = setup.get('shape_param', [-1, 1])
        lo, hi = (sp[0], sp[1]) if isinstance(sp, (list, tuple)) else (-sp, sp)
        self.structure = structure if structure is not None else Structure(funcs, self.d, lo, hi)
        if u_mod.blob is None or v_mod.blob is None:
            raise XnwanError('bind() the networks to the device before building the engine')
        self.theta, self.phi = u_mod.blob, v_mod.blob
        self.Pu, self.Pv = self.theta.data.numel(), self.phi.data.numel()
        z = lambda n: torch.zeros(n, dtype=F64, device=device)  # noqa: E731
        self.adam_u = dict(m=z(self.Pu), v=z(self.Pu), step=torch.zeros(1, dtype=torch.int64, device=device))
        self.adam_v = dict(m=z(self.Pv), v=z(self.Pv), step=torch.zeros(1, dtype=torch.int64, device=device))
        self.grad_u, self.grad_v = z(self.Pu), z(self.Pv)   # the gradient Adam saw in the last sub-step
        self.scal = z(16)

    # ------------------------------------------------------------------------------------------------------------
    # per-sample preparation (once per outer iteration; everything here is parameter-independent)
    # ------------------------------------------------------------------------------------------------------------
    def load_group(self, X, XV, BX, domain, n_glob=None, nb_glob=None):
        """Prepare one group.  The user's callables (h, f, g, func_w, a, b) are evaluated on the device the given
        tensors live on and only their results are uploaded: pass the loader's host tensors to tabulate exactly like
        the reference's CPU path, or device tensors to tabulate on the GPU (float32 transcendental functions then
        differ from the host's in the last bit)."""
        dev, d = self.dev, self.d
        G = Group()
        X, XV = X.detach(), XV.detach()
        BX = BX.detach() if BX is not None else None
        G.domain = domain
        G.N, G.L = X.shape[0], X.shape[1]
        G.Nb = BX.shape[0] if BX is not None else 0
        G.Nglob = float(n_glob if n_glob is not None else G.N)
        G.Nbglob = float(nb_glob if nb_glob is not None else max(G.Nb, 1))
        G.Vol = float(domain.V())
        G.t = X[0, :, 0].to(dev).to(F32).contiguous()
        G.xT = X[:, 0, 1:].to(dev).to(F32).t().contiguous()
        G.xvT = XV[:, 0, 1:].to(dev).to(F32).t().contiguous()
        if XV.shape[1] != G.L:
            raise XnwanError('u- and v-samples of a group must share the time grid')
        # start values and their x-gradient (the h -> y0 path of nabla_x u, src/model.py:95)
        X0 = X[:, 0, :].clone().requires_grad_(True)
        starts_T0 = float(X[0, 0, 0]) == self.setup['T0']
        s = self.funcs['h'](X0) if starts_T0 else self.funcs['g'](X0.unsqueeze(1)).reshape(-1)
        G.start = s.detach().to(dev).to(F64).reshape(-1).contiguous()
        if s.requires_grad:
            G.ghT = torch.autograd.grad(s.sum(), X0)[0][:, 1:].to(dev).to(F64).t().contiguous()
        else:
            G.ghT = torch.zeros(d, G.N, dtype=F64, device=dev)
        G.h = self.funcs['h'](X[:, 0, :]).detach().to(dev).to(F64).reshape(-1).contiguous()
        G.f = _to_LN(self.funcs['f'](X), dev)
        # distance weight on the v-sample and its gradient (nabla phi = w nabla v + v nabla w, src/loss.py:51-63)
        XVl = XV.clone().requires_grad_(True)
        w = domain.func_w(XVl)
        gw = torch.autograd.grad(w.sum(), XVl)[0] if w.requires_grad else torch.zeros_like(XVl)
        if getattr(domain, 'time_independent', False):
            G.w = w[:, 0].detach().to(dev).to(F64).contiguous()
            G.wt = None
        else:
            G.w = _to_LN(w, dev)
            G.wt = _to_LN(gw[:, :, 0], dev)
        G.w0 = w[:, 0].detach().to(dev).to(F64).contiguous()
        G.gwx0T = gw[:, 0, 1:].to(dev).to(F64).t().contiguous()
        if BX is not None:
            G.xbT = BX[:, 0, 1:].to(dev).to(F32).t().contiguous()
            b_T0 = float(BX[0, 0, 0]) == self.setup['T0']
            sb = self.funcs['h'](BX[:, 0, :]) if b_T0 else self.funcs['g'](BX[:, 0, :].unsqueeze(1)).reshape(-1)
            G.start_b = sb.detach().to(dev).to(F64).reshape(-1).contiguous()
            G.g = _to_LN(self.funcs['g'](BX), dev)
            if not torch.equal(BX[0, :, 0].to(dev).to(F32), G.t):
                raise XnwanError('boundary and interior groups of the cube share one time grid')
        G.X = X.to(dev)                          # only read by a general (non-linear) reaction callable c(u, t, x)
        st = self.structure
        G.A0 = G.B0 = None
        if not st.a_identity:
            X1 = X[:, :1, :]
            G.A0 = torch.stack([torch.stack([self.funcs['a'](X1, i, j).to(dev).to(F64)[:, 0] for j in range(d)], 0)
                                for i in range(d)], 0)                       # [d, d, N] at time index 0
        if not st.b_zero:
            X1 = X[:, :1, :]
            G.B0 = torch.stack([self.funcs['b'](X1, i).to(dev).to(F64)[:, 0] for i in range(d)], 0)   # [d, N]
        # work buffers
        e = lambda *s_: torch.empty(*s_, dtype=F64, device=dev)  # noqa: E731
        L, N, Nb, H = G.L, G.N, G.Nb, self.H
        G.u, G.Y, G.v, G.vt = e(L, N), e(L, H, N), e(L, N), e(L, N)
        G.gxv, G.gtv, G.gx, G.gs = e(d, N), e(N), e(d, N), e(N)
        G.ubar, G.vbar, G.s3x = e(L, N), e(L, N), e(N)
        G.ns_u = KN.ode_bwd_slabs(N)
        G.ns_b = KN.ode_bwd_slabs(Nb) if Nb else 0
        G.slab_u = e(G.ns_u + G.ns_b, self.Pu)
        G.slab_v = e(KN.disc_bwd_slabs(N, L), self.Pv)
        if Nb:
            G.ub, G.Yb, G.ubar_b = e(L, Nb), e(L, H, Nb), e(L, Nb)
        return G

    # ------------------------------------------------------------------------------------------------------------
    # shared front half of both sub-steps
    # ------------------------------------------------------------------------------------------------------------
    def _forward(self, G, boundary):
        th, ph = self.theta.data, self.phi.data
        KN.disc_fwd(G.xvT, G.t, ph, self.W, self.q, v=G.v, vt=G.vt)                       # v, dv/dt at all points
        KN.disc_gradx(G.xvT, G.t, ph, self.W, self.q, gxv=G.gxv, gtv=G.gtv)               # nabla_x v at t_0
        KN.ode_fwd(G.xT, G.t, G.start, th, self.method, self.H, self.K, self.m, u=G.u, Y=G.Y)
        if boundary:
            KN.ode_fwd(G.xbT, G.t, G.start_b, th, self.method, self.H, self.K, self.m, u=G.ub, Y=G.Yb)
        # helper backward #1 (src/loss.py:55): G_n = d(sum_l u)/dx_n, including the path through the start value
        KN.ode_bwd(G.xT, G.t, G.start, th, G.Y, None, self.method, self.H, self.K, self.m, want_x=True, want_params=False,
                   gx=G.gx, gs=G.gs)
        Gx = G.gx + G.gs.unsqueeze(0) * G.ghT                                              # [d, N]
        dphi0 = G.w0.unsqueeze(0) * G.gxv + G.v[0].unsqueeze(0) * G.gwx0T                  # nabla_x phi at t_0, [d, N]
        if G.A0 is None:
            s3x = (dphi0 * Gx).sum(0)
        else:
            s3x = torch.einsum('ijn,in,jn->n', G.A0, dphi0, Gx)
        if G.B0 is not None:
            s3x = s3x + G.v[0] * G.w0 * (G.B0 * Gx).sum(0)
        G.s3x.copy_(s3x)
        # reaction term c(u, t, x): linear fast path or the user's callable differentiated by autograd
        G.c = G.cp = None
        ck = self.structure.c_kappa
        if ck is None:
            ul = G.u.t().unsqueeze(2).detach().requires_grad_(True)
            c = self.funcs['c'](G.X, ul)
            cp = torch.autograd.grad(c.sum(), ul)[0] if c.requires_grad else torch.zeros_like(ul)
            G.c, G.cp = _to_LN(c.squeeze(2), self.dev), _to_LN(cp.squeeze(2), self.dev)
            ck = 0.0
        G.ck = ck
        self.scal.zero_()
        KN.weak_partials(G.u, G.v, G.vt, G.w, G.s3x, G.f, G.h, G.Vol, G.Nglob, self.scal, c=G.c, ckappa=ck, wt=G.wt)
        if boundary:
            KN.bdry_partials(G.ub, G.g, self.alpha, G.Nbglob, self.scal, ubar_b=G.ubar_b)
        if self.world is not None:
            self.world.all_reduce(self.scal[0:4])

    def _apply_adam(self, blob, slabs, state, lr, gsum):
        if self.world is None:
            KN.adam(blob.data, slabs, state['m'], state['v'], state['step'], lr, gsum_out=gsum)
        else:
            KN.slab_sum(slabs, out=gsum)
            self.world.all_reduce(gsum)
            KN.adam(blob.data, None, state['m'], state['v'], state['step'], lr, gextra=gsum)

    # ------------------------------------------------------------------------------------------------------------
    def generator_step(self, G):
        """one pass of the generator sub-step body (src/training.py:127-138); returns nothing -- loss in scal[4]"""
        self._forward(G, boundary=G.Nb > 0)
        KN.gen_cotangent(G.u, G.v, G.w, G.h, G.Vol, G.Nglob, G.Nbglob, self.alpha, self.scal, G.ubar, c=G.c, cp=G.cp,
                         ckappa=G.ck, pollution=self.pollution)
        th = self.theta.data
        KN.ode_bwd(G.xT, G.t, G.start, th, G.Y, G.ubar, self.method, self.H, self.K, self.m, want_x=False,
                   want_params=True, gslab=G.slab_u[:G.ns_u])
        if G.Nb:
            KN.ode_bwd(G.xbT, G.t, G.start_b, th, G.Yb, G.ubar_b, self.method, self.H, self.K, self.m, want_x=False,
                       want_params=True, gslab=G.slab_u[G.ns_u:])
        self._apply_adam(self.theta, G.slab_u, self.adam_u, self.config['u_rate'], self.grad_u)

    def discriminator_step(self, G):
        """one pass of the discriminator sub-step body (src/training.py:152-162); loss in scal[5]"""
        self._forward(G, boundary=False)
        KN.disc_cotangent(G.u, G.v, G.w, G.f, G.h, G.Vol, G.Nglob, self.scal, G.vbar, c=G.c, ckappa=G.ck,
                          pollution=self.pollution)
        KN.disc_bwd(G.xvT, G.t, self.phi.data, G.vbar, self.W, self.q, gslab=G.slab_v)
        self._apply_adam(self.phi, G.slab_v, self.adam_v, self.config['v_rate'], self.grad_v)

    # ------------------------------------------------------------------------------------------------------------
    def loss_u(self):
        return self.scal[4]

    def loss_v(self):
        return self.scal[5]

    def predict(self, X):
        """u_theta on a group [N, L, d+1] -> [L, N] (diagnostics; no checkpoints kept)"""
        X = X.detach().to(self.dev)
        starts_T0 = float(X[0, 0, 0]) == self.setup['T0']
        s = self.funcs['h'](X[:, 0, :]) if starts_T0 else self.funcs['g'](X[:, 0, :].unsqueeze(1)).reshape(-1)
        u, _ = KN.ode_fwd(X[:, 0, 1:].to(F32).t().contiguous(), X[0, :, 0].to(F32).contiguous(),
                          s.detach().to(self.dev).to(F64).reshape(-1).contiguous(), self.theta.data, self.method,
                          self.H, self.K, self.m, want_Y=False)
        return u
