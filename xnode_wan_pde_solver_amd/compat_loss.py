"""`loss` with the reference's constructor and method signatures (src/loss.py:12-96) for user code that assembles the
WAN loss itself from `u_net(X)` / `v_net(XV)` outputs.  It runs on torch autograd over the two HIP-backed autograd
Functions of nets.py; the training loop does NOT use it (engine.py computes the same quantities with fused kernels and
never tabulates a[d,d,N,L]).

The reference's side effects are reproduced explicitly rather than by accident:
  * I() runs the two helper backward passes with .backward(), so parameter .grad's receive d(sum u)/dtheta and
    d(sum phi)/dphi (Q1);
  * nabla u / nabla phi are detached copies, the u-factor of the d(phi)/dt term and the phi-factor of the b-term are
    detached (what zeroing X.grad / XV.grad in place does to the saved operands, Q2);
  * X.grad and XV.grad are left zeroed.
"""
import torch


class loss:
    def __init__(self, alpha, a, b, c, h, f, g, setup, domain, device):
        mv = lambda t: t.to(device) if torch.is_tensor(t) else t  # noqa: E731  (tabulated on the host, used on the GPU)
        self.alpha, self.a, self.b, self.c, self.h, self.f, self.g = alpha, mv(a), mv(b), mv(c), mv(h), mv(f), mv(g)
        self.setup, self.T, self.T0 = setup, setup['T'], setup['T0']
        self.func_w, self.V, self.device = domain.func_w, domain.V(), device

    def _input_grad(self, out, weight, X):
        if not X.is_leaf:
            X.retain_grad()
        out.backward(weight, retain_graph=True)
        g = X.grad.detach().clone().to(self.device)
        X.grad.data.zero_()
        return g

    def I(self, y_output_u, y_output_v, X, XV):
        d = self.setup['dim']
        N, L = y_output_u.shape[0], y_output_u.shape[1]
        u, v = y_output_u.squeeze(2), y_output_v.squeeze(2)
        w = self.func_w(XV).to(self.device)
        phi = v * w
        du = self._input_grad(y_output_u, torch.ones_like(y_output_u), X)
        dphi = self._input_grad(phi, torch.ones_like(phi), XV)
        s1 = self.V * (u[:, -1] * v[:, -1] - self.h * v[:, 0]) / N
        s2 = self.V * (u.detach() * dphi[:, :, 0]) / N / L
        s3 = torch.zeros_like(u)
        for i in range(d):
            for j in range(d):
                s3 = s3 + self.a[i, j] * dphi[:, :, i + 1] * du[:, :, j + 1]
            s3 = s3 + self.b[i] * phi.detach() * du[:, :, i + 1]
        s3 = s3 + self.c.squeeze(2) * u * phi + self.f * phi
        return torch.sum(s1 - torch.sum(s2 - (self.V / N / L) * s3, 1), 0)

    def init(self, y_output_u):
        return torch.mean((y_output_u[:, 0] - self.h.unsqueeze(1)) ** 2)

    def bdry(self, u_net, border_data):
        return torch.mean((u_net(border_data) - self.g.unsqueeze(2)) ** 2)

    def int(self, y_output_u, y_output_v, X, XV):
        n_pts = y_output_v.shape[0] * y_output_v.shape[1]
        return torch.log(self.I(y_output_u, y_output_v, X, XV) ** 2) - torch.log(self.V * torch.sum(y_output_v ** 2) / n_pts)

    def u(self, y_output_u, y_output_v, u_net, X, XV, border):
        return self.int(y_output_u, y_output_v, X, XV) + self.alpha * (self.init(y_output_u) + self.bdry(u_net, border))

    def v(self, y_output_u, y_output_v, X, XV):
        return -self.int(y_output_u, y_output_v, X, XV)
