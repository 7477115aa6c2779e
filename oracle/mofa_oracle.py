"""CPU restatement of the MOFA+ variational updates behind mu.tl.mofa (TEST INFRASTRUCTURE).

PARITY UNPINNED.  muon only configures and calls the third-party package mofapy2
(/root/reference/muon/_core/tools.py:418-423 import, :455-527 options, :583-585 build/run);
mofapy2 is not vendored, not version-pinned (pyproject.toml lists a bare `mofapy2`; CI installs
git HEAD) and not installable here, so its source could not be consulted.  This file restates
the *published* MOFA / MOFA+ mean-field updates (Argelaguet et al. 2018, 2020: Gaussian
likelihood, spike-and-slab + ARD prior on the weights, ARD prior on the factors per group,
Gamma noise precision per feature and group) with an explicit, documented initialisation and
schedule.  What the reference's own tests pin at this boundary and this oracle reproduces:
the structural result of tests/test_muon_tools.py:25-44 (5 planted factors recovered out of
10 learnt).  The two 1e-6 golden values of :145-147 depend on mofapy2's private RNG stream
and cannot be reproduced without it.

Everything is dense float64 numpy with explicit loops over factors: clarity over speed.
"""
import numpy as np
from scipy.special import digamma, gammaln

A0 = 1e-14  # Gamma prior shape / rate for tau and alpha (MOFA+ default, uninformative)
B0 = 1e-14
TH_A0 = 1.0  # Beta(1,1) prior on the spike-and-slab inclusion probability
TH_B0 = 1.0
TOL = {"fast": 5e-4, "medium": 5e-5, "slow": 5e-6}  # % change of the ELBO w.r.t. its first value


def prepare_views(views, groups, center_groups=True, scale_views=False, scale_groups=False):
    """What tools.py:283-287 asks mofapy2.process_data to do for gaussian views, statement by
    statement on dense arrays with NaN = missing (mofapy2 is not under /root/reference; this is
    its published preprocessing, mofapy2/core/utils.py process_data, as of 0.7):
        centring:     per group, features minus their nanmean   (center_groups=True)
                      or features minus their nanmean over ALL samples (center_groups=False)
        scale_views:  the view divided by its nanstd (one scalar)
        scale_groups: every group's block divided by its nanstd (one scalar per group)
    in this order.  The per-(view, group) feature means are returned as intercepts
    (tools.py:283-286)."""
    groups = np.asarray(groups)
    G = int(groups.max()) + 1 if groups.size else 1
    out, intercepts = [], []
    for Y in views:
        Y = np.array(Y, dtype=np.float64, copy=True)
        mu = np.zeros((G, Y.shape[1]))
        for g in range(G):
            idx = groups == g
            with np.errstate(invalid="ignore"):
                mu[g] = np.nanmean(Y[idx], axis=0) if idx.any() else 0.0
            mu[g] = np.nan_to_num(mu[g])
        if center_groups:
            for g in range(G):
                Y[groups == g] -= mu[g]
        else:
            with np.errstate(invalid="ignore"):
                Y -= np.nan_to_num(np.nanmean(Y, axis=0))
        if scale_views:
            s = np.nanstd(Y)
            if s > 0:
                Y /= s
        if scale_groups:
            for g in range(G):
                idx = groups == g
                s = np.nanstd(Y[idx]) if idx.any() else 0.0
                if s > 0:
                    Y[idx] /= s
        out.append(Y)
        intercepts.append(mu)
    return out, intercepts


def init_state(N, Ds, G, K, seed):
    """Explicit initialisation shared with the GPU engine: Z ~ N(0,1) from numpy's
    default_rng(seed) (variance 1), W = 0, all Gamma/Beta nodes at expectation 1 / logit 0."""
    rng = np.random.default_rng(seed)
    st = {
        "EZ": rng.standard_normal((N, K)),
        "EW": [np.zeros((D, K)) for D in Ds],
        "EW2": [np.ones((D, K)) for D in Ds],
        "tau": [np.ones((G, D)) for D in Ds],
        "ltau": [np.zeros((G, D)) for D in Ds],
        "alpha_w": [np.ones(K) for _ in Ds],
        "lalpha_w": [np.zeros(K) for _ in Ds],
        "alpha_z": np.ones((G, K)),
        "lalpha_z": np.zeros((G, K)),
        "lth": [np.full(K, digamma(1.0) - digamma(2.0)) for _ in Ds],
        "l1mth": [np.full(K, digamma(1.0) - digamma(2.0)) for _ in Ds],
    }
    st["EZ2"] = st["EZ"] ** 2 + 1.0
    return st


def run(views, groups=None, n_factors=10, n_iterations=1000, convergence_mode="fast", seed=1,
        ard_weights=True, ard_factors=True, spikeslab_weights=True, center_groups=True,
        scale_views=False, scale_groups=False, min_iterations=2, state=None, prepared=False,
        callback=None):
    """Coordinate-ascent VI.  ``views``: list of N x D_m arrays (NaN rows = sample missing in
    that view).  Returns dict with Z, W (list), ELBO trace, R2 per (view, group, factor) in %."""
    M = len(views)
    N = views[0].shape[0]
    groups = np.zeros(N, dtype=np.int64) if groups is None else np.asarray(groups, dtype=np.int64)
    G = int(groups.max()) + 1
    K = int(n_factors)
    if not prepared:
        views, intercepts = prepare_views(views, groups, center_groups, scale_views, scale_groups)
    else:
        intercepts = None
    mask = [~np.isnan(Y).any(axis=1) for Y in views]  # sample present in view m
    for m in range(M):
        if np.isnan(views[m][mask[m]]).any():
            raise NotImplementedError("element-wise missing values are not supported")
    Y0 = [np.where(mask[m][:, None], views[m], 0.0) for m in range(M)]
    Ds = [Y.shape[1] for Y in views]
    st = init_state(N, Ds, G, K, seed) if state is None else state
    gidx = [np.nonzero(groups == g)[0] for g in range(G)]
    Ngm = np.array([[mask[m][gidx[g]].sum() for g in range(G)] for m in range(M)], dtype=np.float64)
    Ng = np.array([len(i) for i in gidx], dtype=np.float64)
    yy = [np.stack([(Y0[m][gidx[g]] ** 2).sum(axis=0) for g in range(G)]) for m in range(M)]

    EZ, EZ2 = st["EZ"], st["EZ2"]
    elbos = []
    gamma = [np.ones((D, K)) for D in Ds]
    EWh2 = [np.ones((D, K)) for D in Ds]
    sig2w = [np.ones((D, K)) for D in Ds]
    muw = [np.zeros((D, K)) for D in Ds]
    sig2z = np.ones((N, K))

    def zstats(m):
        Gz = np.zeros((G, K, K)); Z2 = np.zeros((G, K)); B = np.zeros((G, Ds[m], K))
        for g in range(G):
            i = gidx[g][mask[m][gidx[g]]]
            Gz[g] = EZ[i].T @ EZ[i]
            Z2[g] = EZ2[i].sum(axis=0)
            B[g] = Y0[m][i].T @ EZ[i]
        return Gz, Z2, B

    for it in range(n_iterations):
        # ---- W (per view; Gauss-Seidel over factors) ---------------------------------------
        for m in range(M):
            Gz, Z2, B = zstats(m)
            tau = st["tau"][m]
            EW, EW2 = st["EW"][m], st["EW2"][m]
            aw = st["alpha_w"][m] if ard_weights else np.ones(K)
            for k in range(K):
                t = np.zeros(Ds[m]); q = np.zeros(Ds[m])
                for g in range(G):
                    cross = EW @ Gz[g][:, k] - EW[:, k] * Gz[g][k, k]
                    t += tau[g] * (B[g][:, k] - cross)
                    q += tau[g] * Z2[g][k]
                prec = q + aw[k]
                s2 = 1.0 / prec
                mu = t * s2
                if spikeslab_weights:
                    lam = (st["lth"][m][k] - st["l1mth"][m][k] + 0.5 * np.log(aw[k])
                           - 0.5 * np.log(prec) + 0.5 * t * t * s2)
                    gam = 1.0 / (1.0 + np.exp(-lam))
                else:
                    gam = np.ones(Ds[m])
                EW[:, k] = gam * mu
                EW2[:, k] = gam * (mu * mu + s2)
                gamma[m][:, k] = gam
                EWh2[m][:, k] = gam * (mu * mu + s2) + (1.0 - gam) / aw[k]
                sig2w[m][:, k] = s2
                muw[m][:, k] = mu
        # ---- Z (per sample; Gauss-Seidel over factors) -----------------------------------------
        A = [np.zeros((N, K)) for _ in range(M)]
        Gw = np.zeros((M, G, K, K)); dw2 = np.zeros((M, G, K))
        for m in range(M):
            for g in range(G):
                TW = st["tau"][m][g][:, None] * st["EW"][m]
                i = gidx[g]
                A[m][i] = Y0[m][i] @ TW
                Gw[m, g] = st["EW"][m].T @ TW
                dw2[m, g] = (st["tau"][m][g][:, None] * st["EW2"][m]).sum(axis=0)
        az = st["alpha_z"] if ard_factors else np.ones((G, K))
        for g in range(G):
            i = gidx[g]
            for k in range(K):
                num = np.zeros(len(i)); prec = np.full(len(i), az[g, k])
                for m in range(M):
                    mk = mask[m][i].astype(np.float64)
                    cross = EZ[i] @ Gw[m, g][:, k] - EZ[i, k] * Gw[m, g][k, k]
                    num += mk * (A[m][i, k] - cross)
                    prec += mk * dw2[m, g][k]
                EZ[i, k] = num / prec
                sig2z[i, k] = 1.0 / prec
                EZ2[i, k] = EZ[i, k] ** 2 + 1.0 / prec
        # ---- Tau, Alpha, Theta -------------------------------------------------------------------
        lik = 0.0
        for m in range(M):
            Gz, Z2, B = zstats(m)
            EW, EW2 = st["EW"][m], st["EW2"][m]
            for g in range(G):
                S = (yy[m][g] - 2.0 * (EW * B[g]).sum(axis=1) + ((EW @ Gz[g]) * EW).sum(axis=1)
                     + EW2 @ Z2[g] - (EW ** 2) @ np.diag(Gz[g]))
                a = A0 + 0.5 * Ngm[m, g]
                b = B0 + 0.5 * S
                st["tau"][m][g] = a / b
                st["ltau"][m][g] = digamma(a) - np.log(b)
                lik += np.sum(0.5 * Ngm[m, g] * (st["ltau"][m][g] - np.log(2 * np.pi))
                              - 0.5 * st["tau"][m][g] * S)
                lik += np.sum(_gamma_kl(A0, B0, a, b, st["tau"][m][g], st["ltau"][m][g]))
            if ard_weights:
                a = A0 + 0.5 * Ds[m]
                b = B0 + 0.5 * EWh2[m].sum(axis=0)
                st["alpha_w"][m] = a / b
                st["lalpha_w"][m] = digamma(a) - np.log(b)
            if spikeslab_weights:
                sg = gamma[m].sum(axis=0)
                a = TH_A0 + sg
                b = TH_B0 + Ds[m] - sg
                st["lth"][m] = digamma(a) - digamma(a + b)
                st["l1mth"][m] = digamma(b) - digamma(a + b)
        if ard_factors:
            for g in range(G):
                a = A0 + 0.5 * Ng[g]
                b = B0 + 0.5 * EZ2[gidx[g]].sum(axis=0)
                st["alpha_z"][g] = a / b
                st["lalpha_z"][g] = digamma(a) - np.log(b)
        # ---- ELBO (after the full sweep) -----------------------------------------------------
        elbo = lik
        for m in range(M):
            aw = st["alpha_w"][m] if ard_weights else np.ones(K)
            law = st["lalpha_w"][m] if ard_weights else np.zeros(K)
            gam = gamma[m]
            elbo += np.sum(0.5 * law - 0.5 * aw * EWh2[m])
            elbo += np.sum(gam * 0.5 * np.log(sig2w[m]) + (1 - gam) * 0.5 * np.log(1.0 / aw) + 0.5)
            if spikeslab_weights:
                elbo += np.sum(gam * st["lth"][m] + (1 - gam) * st["l1mth"][m])
                with np.errstate(divide="ignore", invalid="ignore"):
                    ent = -(gam * np.log(gam) + (1 - gam) * np.log1p(-gam))
                elbo += np.sum(np.nan_to_num(ent))
                sg = gam.sum(axis=0)
                a = TH_A0 + sg; b = TH_B0 + Ds[m] - sg
                elbo += np.sum(_beta_kl(TH_A0, TH_B0, a, b, st["lth"][m], st["l1mth"][m]))
            if ard_weights:
                a = A0 + 0.5 * Ds[m]; b = B0 + 0.5 * EWh2[m].sum(axis=0)
                elbo += np.sum(_gamma_kl(A0, B0, a, b, aw, law))
        az = st["alpha_z"] if ard_factors else np.ones((G, K))
        laz = st["lalpha_z"] if ard_factors else np.zeros((G, K))
        for g in range(G):
            i = gidx[g]
            elbo += np.sum(0.5 * laz[g] - 0.5 * az[g] * EZ2[i] + 0.5 * np.log(sig2z[i]) + 0.5)
            if ard_factors:
                a = A0 + 0.5 * Ng[g]; b = B0 + 0.5 * EZ2[i].sum(axis=0)
                elbo += np.sum(_gamma_kl(A0, B0, a, b, az[g], laz[g]))
        elbos.append(float(elbo))
        if callback is not None:
            callback(it, st, elbos)
        if it >= min_iterations and len(elbos) >= 2:
            delta_pct = 100.0 * abs((elbos[-1] - elbos[-2]) / elbos[0])
            if delta_pct < TOL[convergence_mode]:
                break

    r2 = variance_explained(Y0, mask, gidx, EZ, st["EW"])
    return {"Z": EZ, "W": st["EW"], "elbo": elbos, "r2": r2, "intercepts": intercepts,
            "state": st, "iterations": len(elbos)}


def _gamma_kl(a0, b0, a, b, ex, elx):
    """E[ln p(x)] - E[ln q(x)] for Gamma prior (a0,b0) and Gamma posterior (a,b)."""
    lp = a0 * np.log(b0) - gammaln(a0) + (a0 - 1.0) * elx - b0 * ex
    lq = a * np.log(b) - gammaln(a) + (a - 1.0) * elx - b * ex
    return lp - lq


def _beta_kl(a0, b0, a, b, elx, el1mx):
    lb = lambda p, q: gammaln(p) + gammaln(q) - gammaln(p + q)  # noqa: E731
    return (lb(a, b) - lb(a0, b0)) + (a0 - a) * elx + (b0 - b) * el1mx


def variance_explained(Y0, mask, gidx, EZ, EW):
    """R2 (in %) of each factor alone, per view and group (what muon reads back from
    mofapy2's `variance_explained/r2_per_factor`, tools.py:681-697)."""
    M, G, K = len(Y0), len(gidx), EZ.shape[1]
    r2 = np.zeros((M, G, K))
    for m in range(M):
        for g in range(G):
            i = gidx[g][mask[m][gidx[g]]]
            ss = (Y0[m][i] ** 2).sum()
            for k in range(K):
                res = Y0[m][i] - np.outer(EZ[i, k], EW[m][:, k])
                r2[m, g, k] = 100.0 * (1.0 - (res ** 2).sum() / ss) if ss > 0 else 0.0
    return r2


# ------------------------------------------------------------------------------------------------
# Non-gaussian likelihoods and element-wise missing values (PARITY UNPINNED like the
# rest of this file: mofapy2 is not under /root/reference).  The reference reaches them through
# tools.py:272-280 (`guess_likelihoods`) / the `likelihoods` argument (:296) and the NaN handling of
# :144-169.  Restated here from the published bounds (MOFA, Argelaguet et al. 2018, Methods "Non-
# gaussian likelihoods", after Seeger & Bouchard 2012 and Jaakkola & Jordan 2000):
#   every view is a gaussian model on pseudo-data yhat with an element-wise precision Omega,
#     gaussian   Omega_nd = tau_gd m_nd               R_nd := Omega yhat = tau_gd m_nd y_nd
#     poisson    Omega_nd = kappa_d m_nd,  kappa_d = 0.25 + 0.17 max_n y_nd   (rate ln(1 + e^zeta))
#                R_nd = m_nd (kappa_d zeta_nd - sigmoid(zeta_nd) (1 - y_nd / ln(1 + e^zeta_nd)))
#     bernoulli  Omega_nd = 2 lambda(xi_nd) m_nd,  lambda(x) = tanh(x / 2) / (4 x),  xi^2 = E[(w_d z_n)^2]
#                R_nd = m_nd (y_nd - 1/2)
#   with m_nd = 1 where y_nd is observed, zeta = <Z><W>^T.  Expansion points (zeta, xi) are refreshed
#   from the current expectations before the W update and again before the Z update of an iteration
#   (any refresh order is a valid coordinate ascent on the bound).  Non-gaussian views are neither
#   centred nor scaled (process_data touches gaussian views only); tau is a node for gaussian views only.
#   ELBO data terms: gaussian as above (per element), poisson sum m (y ln rate - rate), bernoulli
#   sum m (y zeta - ln(1 + e^zeta)), both at zeta = <Z><W>^T after the sweep.
def _lambda_jj(x):
    x = np.maximum(np.abs(x), 1e-8)
    return np.tanh(0.5 * x) / (4.0 * x)


def _softplus(x):
    return np.logaddexp(0.0, x)


def _sigmoid(x):
    return 0.5 * (1.0 + np.tanh(0.5 * x))


def _omega_r(lik, Y, Mk, EZ, EZ2, EW, EW2, tau_rows, kappa):
    """(Omega, R, zeta) of one view for all samples: dense N x D arrays."""
    zeta = EZ @ EW.T
    if lik == "gaussian":
        Om = tau_rows * Mk
        return Om, Om * Y, zeta
    if lik == "poisson":
        rate = np.maximum(_softplus(zeta), 1e-300)
        Om = kappa[None, :] * Mk
        return Om, Mk * (kappa[None, :] * zeta - _sigmoid(zeta) * (1.0 - Y / rate)), zeta
    xi2 = zeta ** 2 + EZ2 @ EW2.T - (EZ ** 2) @ (EW ** 2).T
    Om = 2.0 * _lambda_jj(np.sqrt(np.maximum(xi2, 0.0))) * Mk
    return Om, Mk * (Y - 0.5), zeta


def run_general(views, likelihoods, groups=None, n_factors=10, n_iterations=1000, convergence_mode="fast",
                seed=1, ard_weights=True, ard_factors=True, spikeslab_weights=True, center_groups=True,
                scale_views=False, scale_groups=False, min_iterations=2, spikeslab_factors=False):
    """Coordinate-ascent VI with per-view likelihoods in {'gaussian', 'poisson', 'bernoulli'} and
    element-wise missing values (NaN).  Dense numpy, loops over factors.

    ``spikeslab_factors`` (/root/reference/muon/_core/tools.py:305,486 -> mofapy2 set_model_options): the factors get
    the spike-and-slab prior the weights have - z_nk = s_nk zhat_nk, s_nk ~ Bernoulli(theta_gk), zhat_nk ~ N(0, 1 /
    alpha_gk), theta_gk ~ Beta - i.e. the W node's update with samples in the place of features and one (alpha, theta)
    pair per (group, factor).  Restated from the MOFA+ publication like the rest of this file: parity unpinned."""
    M, N = len(views), views[0].shape[0]
    groups = np.zeros(N, dtype=np.int64) if groups is None else np.asarray(groups, dtype=np.int64)
    G, K = int(groups.max()) + 1, int(n_factors)
    Ys, masks, kappas, intercepts = [], [], [], []
    for m, (Y, lik) in enumerate(zip(views, likelihoods)):
        Y = np.array(Y, dtype=np.float64, copy=True)
        if lik == "gaussian":
            (Y,), (mu,) = prepare_views([Y], groups, center_groups, scale_views, scale_groups)
        else:
            mu = np.zeros((G, Y.shape[1]))
            for g in range(G):
                with np.errstate(invalid="ignore"):
                    mu[g] = np.nan_to_num(np.nanmean(Y[groups == g], axis=0)) if (groups == g).any() else 0.0
        Mk = (~np.isnan(Y)).astype(np.float64)
        Ys.append(np.nan_to_num(Y))
        masks.append(Mk)
        kappas.append(0.25 + 0.17 * (Ys[-1] * Mk).max(axis=0) if lik == "poisson" else None)
        intercepts.append(mu)
    Ds = [Y.shape[1] for Y in Ys]
    st = init_state(N, Ds, G, K, seed)
    EZ, EZ2 = st["EZ"], st["EZ2"]
    gidx = [np.nonzero(groups == g)[0] for g in range(G)]
    Ng = np.array([len(i) for i in gidx], dtype=np.float64)
    gamma = [np.ones((D, K)) for D in Ds]
    EWh2 = [np.ones((D, K)) for D in Ds]
    sig2w = [np.ones((D, K)) for D in Ds]
    sig2z = np.ones((N, K))
    gamma_z = np.ones((N, K))
    EZh2 = EZ2.copy()
    c0 = digamma(1.0) - digamma(2.0)
    lthz, l1mthz = np.full((G, K), c0), np.full((G, K), c0)
    pres = [(Mk.sum(axis=1) > 0).astype(np.float64) for Mk in masks]  # sample has any entry in view m
    elbos = []

    def om_r(m):
        return _omega_r(likelihoods[m], Ys[m], masks[m], EZ, EZ2, st["EW"][m], st["EW2"][m],
                        st["tau"][m][groups], kappas[m])

    for it in range(n_iterations):
        # ---- W ------------------------------------------------------------------------------------
        for m in range(M):
            Om, R, _ = om_r(m)
            EW, EW2 = st["EW"][m], st["EW2"][m]
            aw = st["alpha_w"][m] if ard_weights else np.ones(K)
            b = R.T @ EZ                                   # D x K
            for k in range(K):
                t = b[:, k].copy()
                for j in range(K):
                    if j != k:
                        t -= EW[:, j] * (Om.T @ (EZ[:, k] * EZ[:, j]))
                prec = Om.T @ EZ2[:, k] + aw[k]
                s2 = 1.0 / prec
                mu = t * s2
                if spikeslab_weights:
                    lam = (st["lth"][m][k] - st["l1mth"][m][k] + 0.5 * np.log(aw[k]) - 0.5 * np.log(prec)
                           + 0.5 * t * t * s2)
                    gam = 1.0 / (1.0 + np.exp(-lam))
                else:
                    gam = np.ones(Ds[m])
                EW[:, k] = gam * mu
                EW2[:, k] = gam * (mu * mu + s2)
                gamma[m][:, k] = gam
                EWh2[m][:, k] = gam * (mu * mu + s2) + (1.0 - gam) / aw[k]
                sig2w[m][:, k] = s2
        # ---- Z ------------------------------------------------------------------------------------
        OR = [om_r(m) for m in range(M)]
        az = (st["alpha_z"] if ard_factors else np.ones((G, K)))[groups]  # N x K
        for k in range(K):
            num = np.zeros(N)
            prec = az[:, k].copy()
            for m in range(M):
                Om, R, _ = OR[m]
                EW, EW2 = st["EW"][m], st["EW2"][m]
                a = R @ EW[:, k]
                for j in range(K):
                    if j != k:
                        a -= EZ[:, j] * (Om @ (EW[:, k] * EW[:, j]))
                num += a
                prec += Om @ EW2[:, k]
            s2 = 1.0 / prec
            mu = num * s2
            if spikeslab_factors:
                lam = (lthz[groups, k] - l1mthz[groups, k] + 0.5 * np.log(az[:, k]) - 0.5 * np.log(prec)
                       + 0.5 * num * num * s2)
                gz = 1.0 / (1.0 + np.exp(-lam))
            else:
                gz = np.ones(N)
            EZ[:, k] = gz * mu
            sig2z[:, k] = s2
            EZ2[:, k] = gz * (mu * mu + s2)
            gamma_z[:, k] = gz
            EZh2[:, k] = gz * (mu * mu + s2) + (1.0 - gz) / az[:, k]
        # ---- tau (gaussian views) and the data terms of the ELBO ----------------------------------------
        lik_sum = 0.0
        for m in range(M):
            EW, EW2 = st["EW"][m], st["EW2"][m]
            zeta = EZ @ EW.T
            Mk, Y = masks[m], Ys[m]
            if likelihoods[m] == "gaussian":
                var = EZ2 @ EW2.T - (EZ ** 2) @ (EW ** 2).T
                res = Mk * ((Y - zeta) ** 2 + var)
                for g in range(G):
                    i = gidx[g]
                    S = res[i].sum(axis=0)
                    Ngd = Mk[i].sum(axis=0)
                    a = A0 + 0.5 * Ngd
                    b = B0 + 0.5 * S
                    st["tau"][m][g] = a / b
                    st["ltau"][m][g] = digamma(a) - np.log(b)
                    lik_sum += np.sum(0.5 * Ngd * (st["ltau"][m][g] - np.log(2 * np.pi)) - 0.5 * st["tau"][m][g] * S)
                    lik_sum += np.sum(_gamma_kl(A0, B0, a, b, st["tau"][m][g], st["ltau"][m][g]))
            elif likelihoods[m] == "poisson":
                rate = np.maximum(_softplus(zeta), 1e-300)
                lik_sum += np.sum(Mk * (Y * np.log(rate) - rate))
            else:
                lik_sum += np.sum(Mk * (Y * zeta - _softplus(zeta)))
            if ard_weights:
                a = A0 + 0.5 * Ds[m]
                b = B0 + 0.5 * EWh2[m].sum(axis=0)
                st["alpha_w"][m] = a / b
                st["lalpha_w"][m] = digamma(a) - np.log(b)
            if spikeslab_weights:
                sg = gamma[m].sum(axis=0)
                a = TH_A0 + sg
                b = TH_B0 + Ds[m] - sg
                st["lth"][m] = digamma(a) - digamma(a + b)
                st["l1mth"][m] = digamma(b) - digamma(a + b)
        if ard_factors:
            for g in range(G):
                a = A0 + 0.5 * Ng[g]
                b = B0 + 0.5 * EZh2[gidx[g]].sum(axis=0)  # (= <z^2> without the spike: gamma = 1)
                st["alpha_z"][g] = a / b
                st["lalpha_z"][g] = digamma(a) - np.log(b)
        if spikeslab_factors:
            for g in range(G):
                sg = gamma_z[gidx[g]].sum(axis=0)
                a = TH_A0 + sg
                b = TH_B0 + Ng[g] - sg
                lthz[g] = digamma(a) - digamma(a + b)
                l1mthz[g] = digamma(b) - digamma(a + b)
        # ---- prior / entropy terms of the ELBO (same as run()) -------------------------------------------
        elbo = lik_sum
        for m in range(M):
            aw = st["alpha_w"][m] if ard_weights else np.ones(K)
            law = st["lalpha_w"][m] if ard_weights else np.zeros(K)
            gam = gamma[m]
            elbo += np.sum(0.5 * law - 0.5 * aw * EWh2[m])
            elbo += np.sum(gam * 0.5 * np.log(sig2w[m]) + (1 - gam) * 0.5 * np.log(1.0 / aw) + 0.5)
            if spikeslab_weights:
                elbo += np.sum(gam * st["lth"][m] + (1 - gam) * st["l1mth"][m])
                with np.errstate(divide="ignore", invalid="ignore"):
                    ent = -(gam * np.log(gam) + (1 - gam) * np.log1p(-gam))
                elbo += np.sum(np.nan_to_num(ent))
                sg = gam.sum(axis=0)
                a = TH_A0 + sg; b = TH_B0 + Ds[m] - sg
                elbo += np.sum(_beta_kl(TH_A0, TH_B0, a, b, st["lth"][m], st["l1mth"][m]))
            if ard_weights:
                a = A0 + 0.5 * Ds[m]; b = B0 + 0.5 * EWh2[m].sum(axis=0)
                elbo += np.sum(_gamma_kl(A0, B0, a, b, aw, law))
        azg = st["alpha_z"] if ard_factors else np.ones((G, K))
        laz = st["lalpha_z"] if ard_factors else np.zeros((G, K))
        for g in range(G):
            i = gidx[g]
            gz = gamma_z[i]
            elbo += np.sum(0.5 * laz[g] - 0.5 * azg[g] * EZh2[i])
            elbo += np.sum(gz * 0.5 * np.log(sig2z[i]) + (1 - gz) * 0.5 * np.log(1.0 / azg[g]) + 0.5)
            if spikeslab_factors:
                elbo += np.sum(gz * lthz[g] + (1 - gz) * l1mthz[g])
                with np.errstate(divide="ignore", invalid="ignore"):
                    ent = -(gz * np.log(gz) + (1 - gz) * np.log1p(-gz))
                elbo += np.sum(np.nan_to_num(ent))
                sg = gz.sum(axis=0)
                a = TH_A0 + sg; b = TH_B0 + Ng[g] - sg
                elbo += np.sum(_beta_kl(TH_A0, TH_B0, a, b, lthz[g], l1mthz[g]))
            if ard_factors:
                a = A0 + 0.5 * Ng[g]; b = B0 + 0.5 * EZh2[i].sum(axis=0)
                elbo += np.sum(_gamma_kl(A0, B0, a, b, azg[g], laz[g]))
        elbos.append(float(elbo))
        if it >= min_iterations and len(elbos) >= 2:
            if 100.0 * abs((elbos[-1] - elbos[-2]) / elbos[0]) < TOL[convergence_mode]:
                break

    # R2 per factor on the (pseudo-)data the last sweep worked with: yhat = R / Omega where observed
    r2 = np.zeros((M, G, K))
    for m in range(M):
        Om, R, _ = om_r(m)
        with np.errstate(divide="ignore", invalid="ignore"):
            Yh = np.where(masks[m] > 0, R / np.where(Om > 0, Om, 1.0), 0.0)
        for g in range(G):
            i = gidx[g]
            ss = (masks[m][i] * Yh[i] ** 2).sum()
            for k in range(K):
                res = masks[m][i] * (Yh[i] - np.outer(EZ[i, k], st["EW"][m][:, k]))
                r2[m, g, k] = 100.0 * (1.0 - (res ** 2).sum() / ss) if ss > 0 else 0.0
    return {"Z": EZ, "W": st["EW"], "elbo": elbos, "r2": r2, "intercepts": intercepts, "state": st,
            "iterations": len(elbos), "pres": pres}
