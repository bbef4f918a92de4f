"""CPU LocalOps for the sharded driver, backed by the numpy oracle.  TEST INFRASTRUCTURE ONLY: lets the real
driver (cglb_amd/distributed.py) run at world_size 2 over gloo without a GPU."""
import math

import numpy as np
import scipy.linalg as sla
import torch

from cglb_amd.distributed import LocalOps
from oracle import cglb_oracle as orc


class OracleLocalOps(LocalOps):
    def __init__(self, kind, X, y, hyp: orc.Hypers, r0, r1):
        self.kind, self.X, self.y, self.hyp = kind, X, y, hyp
        self.device, self.dtype = torch.device("cpu"), torch.float64
        self.N, self.D = X.shape
        self.M = hyp.Z.shape[0]
        self.r0, self.r1 = r0, r1
        self.Kloc = orc.kernel_matrix(kind, X[r0:r1], X, hyp.lengthscales, hyp.variance)  # K_ff[I_g, :]
        self._aat = torch.zeros(self.M * self.M, dtype=torch.float64)

    def set_hypers(self, lengthscales, variance, noise, mean, Z, jitter):
        Z = np.asarray(Z.detach().cpu().numpy() if isinstance(Z, torch.Tensor) else Z, dtype=np.float64).reshape(self.M, self.D)
        self.hyp = orc.Hypers(np.asarray(lengthscales, dtype=np.float64).reshape(-1).copy(), float(variance), float(noise), float(mean), Z.copy(), float(jitter))
        self.Kloc = orc.kernel_matrix(self.kind, self.X[self.r0:self.r1], self.X, self.hyp.lengthscales, self.hyp.variance)
        if hasattr(self, "Kfull"):
            self.Kfull = orc.kernel_matrix(self.kind, self.X, self.X, self.hyp.lengthscales, self.hyp.variance)

    def y_full(self):
        return torch.from_numpy(np.asarray(self.y, dtype=np.float64).copy())

    # common terms
    def setup_local(self):
        h = self.hyp
        kuu = orc.kernel_matrix(self.kind, h.Z, h.Z, h.lengthscales, h.variance) + h.jitter * np.eye(self.M)
        self.L = np.linalg.cholesky(kuu)
        kuf = orc.kernel_matrix(self.kind, h.Z, self.X[self.r0:self.r1], h.lengthscales, h.variance)
        self.A = sla.solve_triangular(self.L, kuf, lower=True) / math.sqrt(h.noise) if self.r1 > self.r0 else np.zeros((self.M, 0))
        self._aat.copy_(torch.from_numpy((self.A @ self.A.T).reshape(-1)))

    def aat_tensor(self):
        return self._aat

    def setup_finish(self):
        self.AAt = self._aat.numpy().reshape(self.M, self.M).copy()
        self.LB = np.linalg.cholesky(self.AAt + np.eye(self.M))
        self.trace = float(np.trace(self.AAt))

    # vectors
    def rhs(self, out): out.copy_(torch.from_numpy(self.y[self.r0:self.r1] - self.hyp.mean))
    def matvec(self, p_full, out):
        p = p_full.numpy()
        out.copy_(torch.from_numpy(self.Kloc @ p + self.hyp.noise * p[self.r0:self.r1]))
    def matvec_dot(self, p_full, out, pdot):
        self.matvec(p_full, out)
        pdot[0] = float(p_full.numpy()[self.r0:self.r1] @ out.numpy())
    def precond_u(self, r, u): u.copy_(torch.from_numpy(self.A @ r.numpy()))
    def _t(self, u):
        return sla.solve_triangular(self.LB.T, sla.solve_triangular(self.LB, u.numpy(), lower=True), lower=False)
    def precond_z(self, r, u, z, rz):
        rp = r.numpy() - self._t(u) @ self.A
        z.copy_(torch.from_numpy(rp / self.hyp.noise))
        if rz is not None:
            rz[0] = float(rp @ r.numpy()) / self.hyp.noise
    def update_v_r(self, v, r, p, Ap, rz, pAp, update_r):
        gamma = float(rz[0]) / float(pAp[0])
        v += gamma * p
        if update_r:
            r -= gamma * Ap
    def residual(self, r, b, Kv): r.copy_(b - Kv)
    def update_p(self, p, z, new_rz, rz, restart):
        if restart:
            p.copy_(z)
        else:
            p.copy_(z + p * (float(new_rz[0]) / float(rz[0])))

    # objective phases (mirror of cglb_api.hip obj_phase1..3)
    def obj_phase1(self, v_full, u):
        v = v_full.numpy()
        self.e = self.y[self.r0:self.r1] - self.hyp.mean
        self.Kv = self.Kloc @ v + self.hyp.noise * v[self.r0:self.r1]
        self.res = self.e - self.Kv
        u.copy_(torch.from_numpy(self.A @ self.res))
    def obj_phase2(self, v_full, u, sc, aw):
        s = self.hyp.noise
        v = v_full.numpy()[self.r0:self.r1]
        rp = self.res - self._t(u) @ self.A
        self.w = rp / s
        uu = self.w + 0.5 * v
        vals = [v @ (self.res + 0.5 * self.Kv), self.w @ self.res, uu @ v, self.w @ self.w, (v + self.w).sum(), uu @ (self.Kv - s * v), 0.0, 0.0]
        sc.copy_(torch.tensor(vals, dtype=torch.float64))
        aw.copy_(torch.from_numpy(self.A @ self.w))
    def obj_phase3(self, v_full, sc, aw, grad):
        h, kind, M, D, N = self.hyp, self.kind, self.M, self.D, self.N
        s, f, ls = h.noise, h.variance, np.asarray(h.lengthscales, dtype=np.float64)
        sigma = math.sqrt(s)
        tau = 1.0 + f / s - self.trace / N
        v = v_full.numpy()
        eye = np.eye(M)
        LBinv = sla.solve_triangular(self.LB, eye, lower=True)
        Binv = LBinv.T @ LBinv
        Linv = sla.solve_triangular(self.L, eye, lower=True)
        c = sigma * (Linv.T @ aw.numpy())
        g_ls, g_Z, g_f, g_s, g_mu = np.zeros(D), np.zeros((M, D)), 0.0, 0.0, 0.0
        Xl = self.X[self.r0:self.r1]
        Xs, Zs, Xls = self.X / ls, h.Z / ls, Xl / ls
        if self.r1 > self.r0:
            Guf = np.outer(c, self.w) + (Linv.T @ ((eye / tau - Binv) @ self.A)) / sigma
            uu = self.w + 0.5 * v[self.r0:self.r1]
            d2 = orc.scaled_sqdist(Xl, self.X, ls)
            Wff = orc.kernel_grad_factor(kind, d2, f) * np.outer(uu, v)
            d2u = orc.scaled_sqdist(h.Z, Xl, ls)
            Wuf = orc.kernel_grad_factor(kind, d2u, f) * Guf
            for d in range(D):
                dl = Xls[:, d][:, None] - Xs[:, d][None, :]
                g_ls[d] += (Wff * dl * dl).sum() / ls[d]
                du = Zs[:, d][:, None] - Xls[:, d][None, :]
                g_ls[d] += (Wuf * du * du).sum() / ls[d]
                g_Z[:, d] += -(Wuf * du).sum(axis=1) / ls[d]
            g_f += (Guf * orc.kernel_from_sqdist(kind, d2u, f)).sum() / f
        if self.r0 == 0:  # replicated terms are added once
            scn = sc.numpy()
            inner = 0.5 * (eye - Binv) - (0.5 / tau) * self.AAt
            Guu = -0.5 * np.outer(c, c) + Linv.T @ inner @ Linv
            d2 = orc.scaled_sqdist(h.Z, h.Z, ls)
            Wuu = orc.kernel_grad_factor(kind, d2, f) * Guu
            for d in range(D):
                dz = Zs[:, d][:, None] - Zs[:, d][None, :]
                g_ls[d] += (Wuu * dz * dz).sum() / ls[d]
                g_Z[:, d] += -2.0 * (Wuu * dz).sum(axis=1) / ls[d]
            g_f += (Guu * orc.kernel_from_sqdist(kind, d2, f)).sum() / f
            g_f += scn[5] / f - N / (2.0 * tau * s)
            g_s += scn[2] + 0.5 * scn[3] + (M - np.trace(Binv)) / (2.0 * s) - N / (2.0 * s) + N * f / (2.0 * tau * s * s) - self.trace / (2.0 * tau * s)
            g_mu += scn[4]
        grad.copy_(torch.from_numpy(np.concatenate([g_ls, [g_f, g_s, g_mu], g_Z.reshape(-1)])))
    def obj_finish(self, sc):
        h, N = self.hyp, self.N
        scn = sc.numpy()
        tau = 1.0 + h.variance / h.noise - self.trace / N
        logdet = -float(np.log(np.diag(self.LB)).sum()) - 0.5 * N * math.log(h.noise) - 0.5 * N * math.log(tau)
        lower, upper = scn[0], scn[0] + 0.5 * scn[1]
        return -upper + logdet - 0.5 * N * math.log(2 * math.pi), lower, upper, logdet


from cglb_amd.distributed import SymLocalOps


class OracleSymLocalOps(OracleLocalOps, SymLocalOps):
    """Oracle-backed local ops of the cyclic-symmetric driver.  The K_ff work is split exactly like the HIP kernels do it:
    256-row blocks of the global upper triangle, block rb belongs to rank rb % world; a block contributes its rows against
    all columns from its own first row on, and the transposed contribution to the columns right of the block."""

    RB = 256
    GB = 512  # row-block size of the gradient N^2 pass (256 threads x 2 rows)

    def __init__(self, kind, X, y, hyp, r0, r1):
        super().__init__(kind, X, y, hyp, r0, r1)
        self.world, self.rank = 1, 0
        self.Kfull = orc.kernel_matrix(kind, X, X, hyp.lengthscales, hyp.variance)

    @property
    def noise(self):
        return self.hyp.noise

    def set_parallel(self, world, rank):
        self.world, self.rank = world, rank

    def rhs_full(self, out): out.copy_(torch.from_numpy(self.y - self.hyp.mean))

    def matvec_cyclic(self, p_full, out):
        p = p_full.numpy()
        res = np.zeros(self.N)
        nrb = (self.N + self.RB - 1) // self.RB
        for rb in range(self.rank, nrb, self.world):
            a, b = rb * self.RB, min((rb + 1) * self.RB, self.N)
            res[a:b] += self.Kfull[a:b, a:] @ p[a:]          # rows of the block against columns >= its first row
            if b < self.N:
                res[b:] += self.Kfull[a:b, b:].T @ p[a:b]    # transposed use of the strictly-right part
        if self.rank == 0:
            res += self.hyp.noise * p                        # the noise term travels with rank 0's partial
        out.copy_(torch.from_numpy(res))

    def vec_dot(self, n, a, b, out): out[0] = float(a.numpy()[:n] @ b.numpy()[:n])
    def vec_update_v_r(self, n, v, r, p, Ap, rz, pAp, update_r): self.update_v_r(v[:n], r[:n], p[:n], Ap[:n], rz, pAp, update_r)
    def vec_residual(self, n, r, b, Kv): r[:n].copy_(b[:n] - Kv[:n])
    def vec_update_p(self, n, p, z, new_rz, rz, restart): self.update_p(p[:n], z[:n], new_rz, rz, restart)
    def vec_axpy(self, n, alpha, x, y): y[:n] += alpha * x[:n]

    def precond_z_seg(self, r_local, u, z_slot, per):
        nloc = self.r1 - self.r0
        part = torch.zeros(1, dtype=torch.float64)
        self.precond_z(r_local, u, z_slot[:nloc], part)
        z_slot[per] = part[0]                                # the slice's extra element: this rank's partial of r^T z

    def vec_update_p_seg(self, n, per, world, p, zseg, new_rz, rz, restart):
        seg = zseg.numpy().reshape(world, per + 1)
        nrz = 0.0
        for g in range(world):                               # rank order, like the kernel
            nrz += float(seg[g, per])
        z = torch.from_numpy(seg[:, :per].reshape(-1)[:n].copy())
        if restart:
            p[:n].copy_(z)
        else:
            p[:n].copy_(z + p[:n] * (nrz / float(rz[0])))
        new_rz[0] = nrz

    def obj_phase1_kv(self, Kv_local, u):
        self.e = self.y[self.r0:self.r1] - self.hyp.mean
        self.Kv = Kv_local.numpy().copy()
        self.res = self.e - self.Kv
        u.copy_(torch.from_numpy(self.A @ self.res))

    def obj_w(self, out): out.copy_(torch.from_numpy(self.w))

    def obj_phase3_cyclic(self, v_full, u_full, sc, aw, grad):
        # everything except the N^2 form exactly as the row-sharded phase 3 ...
        saved = self.Kloc
        super().obj_phase3(v_full, sc, aw, grad)
        g = grad.numpy().copy()
        D, ls, f = self.D, np.asarray(self.hyp.lengthscales, dtype=np.float64), self.hyp.variance
        v, u = v_full.numpy(), u_full.numpy()
        Xs = self.X / ls
        # ... minus the row-sharded N^2 part it added, plus this rank's cyclic share of the symmetric form
        if self.r1 > self.r0:
            Xl = self.X[self.r0:self.r1]
            uu = self.w + 0.5 * v[self.r0:self.r1]
            Wff = orc.kernel_grad_factor(self.kind, orc.scaled_sqdist(Xl, self.X, ls), f) * np.outer(uu, v)
            for d in range(D):
                dl = Xl[:, d][:, None] / ls[d] - Xs[:, d][None, :]
                g[d] -= (Wff * dl * dl).sum() / ls[d]
        nb = (self.N + self.GB - 1) // self.GB
        for rb in range(self.rank, nb, self.world):
            a, b = rb * self.GB, min((rb + 1) * self.GB, self.N)
            h = orc.kernel_grad_factor(self.kind, orc.scaled_sqdist(self.X[a:b], self.X[a:], ls), f)
            W = h * np.outer(u[a:b], v[a:])
            if b < self.N:
                W[:, b - a:] += h[:, b - a:] * np.outer(v[a:b], u[b:])
            for d in range(D):
                dl = Xs[a:b, d][:, None] - Xs[a:, d][None, :]
                g[d] += (W * dl * dl).sum() / ls[d]
        grad.copy_(torch.from_numpy(g))

    # prediction pieces (PredictCG.forward, models.py:334-352)
    def predict_u(self, Kv_local, u):
        res = (self.y[self.r0:self.r1] - self.hyp.mean) - Kv_local.numpy()
        u.copy_(torch.from_numpy(self.A @ res))

    def predict_rows(self, v_full, u, xnew):
        h = self.hyp
        xn = np.asarray(xnew.numpy() if isinstance(xnew, torch.Tensor) else xnew, dtype=np.float64).reshape(-1, self.D)
        if xn.shape[0] == 0:
            return torch.zeros(0, dtype=torch.float64), torch.zeros(0, dtype=torch.float64)
        cg_mean = orc.kernel_matrix(self.kind, xn, self.X, h.lengthscales, h.variance) @ v_full.numpy()      # :334
        kus = orc.kernel_matrix(self.kind, h.Z, xn, h.lengthscales, h.variance)                              # :337
        c = sla.solve_triangular(self.LB, u.numpy(), lower=True) / math.sqrt(h.noise)                        # :343
        tmp1 = sla.solve_triangular(self.L, kus, lower=True)                                                 # :344
        tmp2 = sla.solve_triangular(self.LB, tmp1, lower=True)                                               # :345
        mean = tmp2.T @ c + cg_mean + h.mean                                                                 # :347-348
        var = orc.kernel_diag(self.kind, xn, h.variance) + (tmp2 ** 2).sum(0) - (tmp1 ** 2).sum(0)           # :350-351
        return torch.from_numpy(mean), torch.from_numpy(var)
