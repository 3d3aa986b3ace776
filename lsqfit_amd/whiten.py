"""Whitening setup: the product's mirror of what ``nonlinear_fit`` obtains from
``gvar.PDF`` (src/lsqfit/__init__.py:1892-1900, read back at :553-561,:574,:723 and
src/lsqfit/_utilities.pyx:58-61).

Runs once per fit.  The host keeps what is bookkeeping in the reference too -- the
block search, the svdcut policy, ``nmod`` / ``nblocks`` / ``nchiv`` -- and the
O(B^3) factorisations run ON THE DEVICE (``lsqamd_whiten_blocks``: batched
Cholesky of the correlation matrices, triangular inverses, log-determinants, the
test whether the svdcut floor can bind; the weights stay in HBM and go straight
into ``lsqamd_set_data`` / ``lsqamd_set_prior``).  Only a block whose floor DOES
bind (or that is not positive definite) takes the eigen-mode route of
doc/source/overview.rst:1546-1606 in LAPACK on the host -- which is what gvar does
for every block.  Without a visible GPU (the host-logic tests) everything takes the
LAPACK route; nothing can be fitted there anyway.  It produces what the device path
consumes through ``lsqamd_set_data`` / ``lsqamd_set_prior``:

  * ``wdiag``   1/sdev for the 1x1 rows;
  * per correlated data block the TRANSPOSED whitening matrix ``Wt`` with
    ``W^T W = inv(C_block regulated)``.  When the svdcut floor touches no mode
    (the usual case) ``W = inv(chol(C))`` -- triangular, so the device GEMM
    skips half the work; otherwise the eigen form ``w_i = v_i^T D^-1/sqrt(lam_i)``
    of doc/source/overview.rst:1546-1556 (floor) / :1595-1603 (drop) is used.
    Any such W gives the same chi2, J^T J, J^T f, p, cov, logGBF; only the
    rotation of ``fit.f`` / ``fit.J`` rows inside a block differs from gvar's;
  * the prior precision ``inv(C_prior regulated)`` (diagonal or dense);
  * ``logdet`` (= log det of the regulated covariance), ``nmod`` (svdn),
    ``nblocks``, ``nchiv``.

Blocks are the connected components of the covariance pattern, as gvar's; components that
interleave are made contiguous by a row permutation (``Whitening.perm``) that DeviceProblem applies
to x and un-does in its row-ordered outputs.  Data-prior cross-correlations
(examples/y-noerr.py) go through :func:`joint_whitening`: concat(y, prior) is
whitened as ONE vector, exactly as the reference does, and the prior entries
travel to the device as extra "parameter rows" (lsqamd_set_param_rows).
"""
import numpy as np
import scipy.linalg as sla


def _regulate_block(cov, svdcut, force_eig=False):
    """-> dict(Wt[B,B], modes, tri, logdet, nmod, cov_reg_diag)."""
    cov = np.asarray(cov, float)
    B = cov.shape[0]
    sd = np.sqrt(np.diag(cov))
    if not np.all(np.isfinite(sd)) or np.any(sd <= 0):
        raise ValueError('covariance block has a non-positive variance')
    corr = cov / np.outer(sd, sd)
    cut = 0.0 if svdcut is None else float(svdcut)
    touched = True
    L = None
    if not force_eig:
        try:
            L = sla.cholesky(corr, lower=True)
        except sla.LinAlgError:
            L = None
        if L is not None:
            if cut == 0.0:
                touched = False
            else:
                lmax, lmin = _extreme_eigs(corr, L)
                # generous margin: power-iteration estimates are good to a few percent
                touched = not (lmin > 4.0 * abs(cut) * lmax)
    if not touched:
        Linv = sla.solve_triangular(L, np.eye(B), lower=True)
        W = Linv / sd[None, :]                      # inv(chol(C)) = inv(L) D^-1
        logdet = 2.0 * float(np.sum(np.log(np.diag(L)))) + 2.0 * float(np.sum(np.log(sd)))
        return dict(Wt=np.ascontiguousarray(W.T), modes=B, tri=1, logdet=logdet, nmod=0,
                    var_reg=sd ** 2, S=sd[:, None] * L)
    lam, vec = np.linalg.eigh(corr)
    keep = np.ones(B, bool)
    lam_reg = lam.copy()
    nmod = 0
    if cut != 0.0:
        lmin = abs(cut) * lam[-1]
        low = lam < lmin
        nmod = int(np.sum(low))
        if cut > 0:
            lam_reg[low] = lmin
        else:
            keep = ~low
    if np.any(lam_reg[keep] <= 0):
        raise ValueError('covariance block is not positive definite; use an svdcut')
    lam_k, vec_k = lam_reg[keep], vec[:, keep]
    W = (vec_k / np.sqrt(lam_k)).T / sd[None, :]
    m = W.shape[0]
    Wt = np.zeros((B, B))
    Wt[:, :m] = W.T
    logdet = float(np.sum(np.log(lam_k)) + 2.0 * np.sum(np.log(sd)))
    var_reg = np.einsum('ik,k,ik->i', vec_k, lam_k, vec_k) * sd ** 2 if np.all(keep) else sd ** 2
    # S S^T = regulated block covariance (restricted to the kept modes): sampling factor
    S = sd[:, None] * (vec_k * np.sqrt(lam_k))
    out = dict(Wt=Wt, modes=m, tri=0, logdet=logdet, nmod=nmod, var_reg=var_reg, S=S)
    if cut > 0 and nmod:
        # what the floor ADDED to the covariance: sum over the raised modes of (floor - lam) D v v^T D;
        # its square root drives ``noise=True`` (src/lsqfit/__init__.py:247-256)
        low = lam < abs(cut) * lam[-1]
        out['S_add'] = sd[:, None] * (vec[:, low] * np.sqrt(lam_reg[low] - np.minimum(lam[low], lam_reg[low])))
    return out


class _Block(dict):
    """One covariance block.  ``Wt`` (host copy of the transposed weights) and ``S`` (sampling
    factor, S S^T = regulated covariance) are made on first use: the device path needs neither."""

    @staticmethod
    def _to_host(t):
        """device tensor -> numpy, copied on the package's side stream (not the legacy default stream: _lib.side_stream)"""
        import torch
        from . import _lib
        side = _lib.side_stream(t.device)
        with torch.cuda.stream(side):
            h = t.cpu()
        side.synchronize()
        return h.numpy()

    def __missing__(self, key):
        if key == 'Wt':
            v = self._to_host(self['Wt_dev'])
        elif key == 'S':
            cov = np.asarray(self['cov'], float)
            sd = np.sqrt(np.diag(cov))
            v = sd[:, None] * sla.cholesky(cov / np.outer(sd, sd), lower=True)
        elif key == 'prec':
            v = self._to_host(self['prec_dev'])
        else:
            raise KeyError(key)
        self[key] = v
        return v


def device_available():
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


def _device_regulate(covs, svdcut, want_prec):
    """Same-size covariance blocks -> [_Block or None] through ``lsqamd_whiten_blocks``; None marks
    a block the device hands back (svdcut floor may bind / not positive definite)."""
    import ctypes as C
    import torch
    from . import _lib
    lib = _lib.load()
    nb, B = len(covs), covs[0].shape[0]
    stack = np.ascontiguousarray(covs[0] if nb == 1 else np.stack(covs), np.float64)
    dev = torch.device('cuda', torch.cuda.current_device())
    side = _lib.side_stream(dev)          # the package's own non-blocking stream (never the legacy default stream)
    with torch.cuda.stream(side):
        wt = torch.empty((nb, B, B), dtype=torch.float64, device=dev)
        prec = torch.empty((nb, B, B), dtype=torch.float64, device=dev) if want_prec else None
        nbytes = lib.lsqamd_whiten_work_bytes(B, nb)
        work = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        logdet, lmin, lmax = np.empty(nb), np.empty(nb), np.empty(nb)
        status = np.empty(nb, np.int32)
        rc = lib.lsqamd_whiten_blocks(
            C.c_void_p(side.cuda_stream), B, nb, C.c_void_p(stack.ctypes.data),
            0.0 if svdcut is None else float(svdcut), C.c_void_p(wt.data_ptr()),
            C.c_void_p(prec.data_ptr()) if want_prec else None, C.c_void_p(work.data_ptr()), nbytes,
            _lib.dptr(logdet), _lib.dptr(lmin), _lib.dptr(lmax), status.ctypes.data_as(C.POINTER(C.c_int32)))
        side.synchronize()                # (the call hands host results back, so it has waited already: this is for the record)
    if rc != 0:
        raise RuntimeError('lsqfit_amd: lsqamd_whiten_blocks failed (%s)' % _lib.ERRORS.get(rc, rc))
    del work
    out = []
    for b in range(nb):
        if status[b] != 0:
            out.append(None)
            continue
        k = _Block(Wt_dev=wt[b], modes=B, tri=1, logdet=float(logdet[b]), nmod=0,
                   var_reg=np.diag(covs[b]).copy(), cov=covs[b], lam_bounds=(float(lmin[b]), float(lmax[b])))
        if want_prec:
            k['prec_dev'] = prec[b]
        out.append(k)
    return out


def regulate_blocks(covs, svdcut, want_prec=False, engine=None):
    """[cov] -> [_Block]: the factorisations of every correlated block, on the device when one is
    visible (``engine`` None), grouped by block size; LAPACK eigen route for what the device hands
    back and for ``engine='host'``."""
    if engine is None:
        engine = 'device' if device_available() else 'host'
    if engine not in ('device', 'host', 'eig'):
        raise ValueError("engine must be 'device', 'host' or 'eig'")
    out = [None] * len(covs)
    if engine == 'device':
        by_size = {}
        for i, c in enumerate(covs):
            by_size.setdefault(c.shape[0], []).append(i)
        for B, idx in by_size.items():
            for i, k in zip(idx, _device_regulate([np.asarray(covs[i], float) for i in idx], svdcut, want_prec)):
                out[i] = k
    for i, c in enumerate(covs):
        if out[i] is None:
            # 'eig': every block through its eigen-modes, the basis gvar whitens in -- needed where the fit is NOT invariant under
            # a rotation of a block's whitened rows (robust losses, rows_whitening)
            out[i] = _Block(_regulate_block(c, svdcut, force_eig=(engine in ('device', 'eig'))))
            out[i]['cov'] = np.asarray(c, float)
    return out


def _extreme_eigs(corr, L, iters=40):
    """(lambda_max, lambda_min) of an SPD matrix: exact for small blocks, power /
    inverse-power iteration through its Cholesky factor for large ones."""
    B = corr.shape[0]
    if B <= 1536:
        w = sla.eigvalsh(corr)
        return float(w[-1]), float(w[0])
    rng = np.random.default_rng(12345)
    v = rng.standard_normal(B)
    v /= np.linalg.norm(v)
    lmax = 1.0
    for _ in range(iters):
        w = corr @ v
        lmax = float(np.linalg.norm(w))
        v = w / lmax
    u = rng.standard_normal(B)
    u /= np.linalg.norm(u)
    inv_l = 1.0
    for _ in range(iters):
        w = sla.cho_solve((L, True), u)
        inv_l = float(np.linalg.norm(w))
        u = w / inv_l
    return lmax, 1.0 / inv_l


def _as_blocks(err, n):
    """Normalise an error spec to (sdev[n], [(row0, cov_block), ...], perm): ``perm`` is None, or the
    order (new position -> original index) in which sdev and the block offsets are given.

    err: sdev vector | dense cov (n x n, split at zero off-diagonal bands) |
         dict(sdev=..., blocks=[(row0, cov), ...])."""
    if isinstance(err, dict):
        sd = np.array(err['sdev'], float).reshape(-1)
        blocks = [(int(r0), np.asarray(c, float)) for r0, c in err.get('blocks', [])]
        return sd, blocks, None
    err = np.asarray(err, float)
    if err.ndim <= 1:
        return err.reshape(-1) * np.ones(n), [], None
    if err.shape != (n, n):
        raise ValueError('covariance must be %d x %d' % (n, n))
    sd = np.sqrt(np.diag(err))
    # gvar.evalcov_blocks: one block per connected component of the off-diagonal pattern
    # (tests/test_lsqfit.py:1011-1012,1047-1050).  Components that interleave (rows 0 and 5 correlated,
    # rows 1-4 not) are made contiguous by a permutation the caller applies to its rows.
    comps = _components(err)
    if all(int(c[-1]) - int(c[0]) + 1 == c.size for c in comps):
        return sd, [(int(c[0]), err[c[0]:c[-1] + 1, c[0]:c[-1] + 1]) for c in comps if c.size > 1], None
    perm = np.concatenate(comps)
    blocks, r0 = [], 0
    for c in comps:
        if c.size > 1:
            blocks.append((r0, err[np.ix_(c, c)]))
        r0 += c.size
    return sd[perm], blocks, perm


def _eps_shift(cov, eps):
    """gvar.regulate's ``eps`` mode on one correlated block: (C + eps ||corr||_inf D^2, sqrt of what was
    added as a diagonal factor)."""
    cov = np.asarray(cov, float)
    sd = np.sqrt(np.diag(cov))
    if not np.all(np.isfinite(sd)) or np.any(sd <= 0):
        raise ValueError('covariance block has a non-positive variance')
    norm = float(np.max(np.sum(np.abs(cov / np.outer(sd, sd)), axis=1)))
    out = cov.copy()
    out[np.diag_indices_from(out)] += eps * norm * sd ** 2
    return out, np.sqrt(eps * norm) * sd


class Whitening:
    """Everything the device path needs to know about concat(y, prior)."""

    def __init__(self, ymean, yerr, prior_mean=None, prior_err=None, svdcut=1e-12, eps=None,
                 udata=False, engine=None, noise=False, rng=None):
        """``eps`` (used only when ``svdcut is None``, src/lsqfit/__init__.py:240-245): gvar.regulate's
        other mode -- every correlated block's correlation matrix gets ``eps * ||corr||_inf`` on its
        diagonal, i.e. ``C -> C + eps ||corr||_inf D^2``; no eigen-decomposition is ever needed, so every
        block takes the device's Cholesky route.  Restated from gvar's documentation; the reference holds
        no expected value for it (parity unpinned).  ``noise`` (bool or (data, prior),
        :247-256): means shifted by a draw from what the regulation added (data) / from the prior itself
        (prior), drawn from ``rng`` (numpy Generator or seed) -- not gvar's random stream."""
        if svdcut is not None:
            eps = None
        if eps is not None and eps < 0:
            raise ValueError('eps must not be negative')
        self.engine = engine
        self.svdcut = svdcut
        self.eps = eps
        self.noise = (bool(noise), bool(noise)) if np.ndim(noise) == 0 else (bool(noise[0]), bool(noise[1]))
        added = []                            # (row0, factor): S S^T = what the regulation added to the block
        self.ymean = np.array(ymean, float).reshape(-1)
        N = self.ymean.size
        ysd, yblocks, self.perm = _as_blocks(yerr, N)
        if udata:
            yblocks = []                      # __init__.py:1892-1893
            if self.perm is not None:
                ysd, self.perm = ysd[np.argsort(self.perm)], None
        if self.perm is not None:             # interleaved components: rows reordered, device and all
            self.ymean = self.ymean[self.perm]
        self.logdet = 0.0
        self.nmod = 0
        self.nblocks = {}
        self.wdiag = np.ones(N)
        self.blocks = []                      # dicts: row0, size, modes, tri, Wt
        in_block = np.zeros(N, bool)
        yblocks = sorted(yblocks, key=lambda b: b[0])
        if eps:
            shifted = []
            for r0, cov in yblocks:
                creg, extra = _eps_shift(cov, eps)
                shifted.append((r0, creg))
                added.append((int(r0), extra))
            yblocks = shifted
        for (r0, cov), reg in zip(yblocks, regulate_blocks([c for _, c in yblocks], 0.0 if eps else svdcut,
                                                           engine=engine)):
            B = cov.shape[0]
            reg.update(row0=int(r0), size=int(B))
            self.blocks.append(reg)
            in_block[r0:r0 + B] = True
            self.logdet += reg['logdet']
            self.nmod += B if eps else reg['nmod']
            self.nblocks[B] = self.nblocks.get(B, 0) + 1
            if 'S_add' in reg:
                added.append((int(r0), reg['S_add']))
        if self.noise[0] and added:
            gen = np.random.default_rng(rng)
            for r0, S in added:               # S: [B] (diagonal addition) or [B, m]
                z = gen.standard_normal(S.shape[-1])
                self.ymean[r0:r0 + S.shape[0]] += S * z if S.ndim == 1 else S @ z
        d = ~in_block
        if np.any(ysd[d] <= 0) or not np.all(np.isfinite(ysd[d])):
            raise ValueError('some input data have zero or non-finite standard deviations')
        self.wdiag[d] = 1.0 / ysd[d]
        self.logdet += 2.0 * float(np.sum(np.log(ysd[d])))
        n1 = int(np.sum(d))
        self.n_data = N
        self.logdet_data = self.logdet          # log det of the regulated DATA covariance alone
        self.nchiv_data = n1 + sum(b['modes'] for b in self.blocks)
        # ---- prior
        self.has_prior = prior_mean is not None
        self.prior_mean = None
        self._prior_prec = self.prior_prec_dev = None
        self.prior_dense = False
        self.prior_blocks = []                # dense prior blocks (dicts like self.blocks, row0 = first parameter)
        self._prior_diag = None               # (indices, sdev) of the uncorrelated prior entries
        nprior = 0
        if self.has_prior:
            pm = np.array(prior_mean, float).reshape(-1)
            P = pm.size
            psd, pblocks, pperm = _as_blocks(prior_err, P)
            pidx = np.arange(P) if pperm is None else pperm      # position in (psd, pblocks) -> parameter
            if pperm is not None:
                psd_full = np.empty(P)
                psd_full[pperm] = psd
            else:
                psd_full = psd
            self.prior_mean = pm
            psd = psd_full
            if not pblocks:
                if np.any(psd <= 0):
                    raise ValueError('some priors have zero standard deviations')
                self._prior_prec = 1.0 / psd ** 2
                self._prior_diag = (np.arange(P), psd)
                self.logdet += 2.0 * float(np.sum(np.log(psd)))
                n1 += P
                nprior = P
            else:
                self.prior_dense = True
                pblocks = sorted(pblocks, key=lambda b: b[0])
                if eps:
                    pblocks = [(r0, _eps_shift(cov, eps)[0]) for r0, cov in pblocks]
                pin = np.zeros(P, bool)
                for (r0, cov), reg in zip(pblocks, regulate_blocks([c for _, c in pblocks], 0.0 if eps else svdcut,
                                                                   want_prec=True, engine=engine)):
                    B = cov.shape[0]
                    reg.update(row0=int(r0), size=int(B), idx=pidx[r0:r0 + B])
                    self.prior_blocks.append(reg)
                    pin[reg['idx']] = True
                    self.logdet += reg['logdet']
                    self.nmod += B if eps else reg['nmod']
                    self.nblocks[B] = self.nblocks.get(B, 0) + 1
                    nprior += reg['modes']
                dd = np.nonzero(~pin)[0]
                if np.any(psd[dd] <= 0):
                    raise ValueError('some priors have zero standard deviations')
                self._prior_diag = (dd, psd[dd])
                self.logdet += 2.0 * float(np.sum(np.log(psd[dd])))
                n1 += dd.size
                nprior += dd.size
                k0 = self.prior_blocks[0]
                if len(self.prior_blocks) == 1 and k0['size'] == P and 'prec_dev' in k0 and pperm is None:
                    self.prior_prec_dev = k0['prec_dev']      # one block over all parameters: stays in HBM
            self.prior_sdev = psd
        if n1:
            self.nblocks[1] = n1
        self.nchiv = self.nchiv_data + nprior
        if self.noise[1] and self.has_prior:  # prior means move by a draw from the (regulated) prior
            gen = np.random.default_rng(None if rng is None else np.random.default_rng(rng).integers(1 << 62))
            for idx, S in self.prior_S:
                z = gen.standard_normal(S.shape[-1])
                self.prior_mean[idx] += S * z if np.ndim(S) == 1 else S @ z

    # -- prior pieces the host needs only now and then (made on first use) ---------------------
    @property
    def prior_prec(self):
        """inv(C_prior regulated): P entries (diagonal prior) or P x P."""
        if self._prior_prec is None and self.has_prior:
            P = self.prior_mean.size
            prec = np.zeros((P, P))
            for k in self.prior_blocks:
                ix = np.ix_(k['idx'], k['idx'])
                if 'prec_dev' in k:
                    prec[ix] = k['prec']
                else:
                    W = k['Wt'].T[:k['modes']]
                    prec[ix] = W.T @ W
            dd, sd = self._prior_diag
            prec[dd, dd] = 1.0 / sd ** 2
            self._prior_prec = prec
        return self._prior_prec

    @prior_prec.setter
    def prior_prec(self, value):
        self._prior_prec, self.prior_prec_dev = value, None

    @property
    def prior_W(self):
        """Rows of the whitened prior residual (for fit.f / fit.J): ('diag', w) or
        ('dense', rows of the uncorrelated entries, [rows of each block])."""
        if not self.has_prior:
            return None
        dd, sd = self._prior_diag
        if not self.prior_dense:
            return ('diag', 1.0 / sd)
        P = self.prior_mean.size
        diag_rows = np.zeros((dd.size, P))
        diag_rows[np.arange(dd.size), dd] = 1.0 / sd
        Wrows = []
        for k in self.prior_blocks:
            W = k['Wt'].T[:k['modes']]
            full = np.zeros((W.shape[0], P))
            full[:, k['idx']] = W
            Wrows.append(full)
        return ('dense', diag_rows, Wrows)

    @property
    def prior_S(self):
        """[(indices, sampling factor)]: S S^T = regulated prior covariance."""
        out = [(k['idx'], k['S']) for k in self.prior_blocks]
        dd, sd = self._prior_diag
        if dd.size:
            out.append((dd, sd))
        return out

    # -- Gaussian draws with the regulated covariance (what gvar.bootstrap_iter produces) -----
    def draw_data(self, rng, n):
        """(n, N) deviates with the regulated data covariance."""
        out = rng.standard_normal((n, self.n_data)) / self.wdiag
        for k in self.blocks:
            S = k['S']
            out[:, k['row0']:k['row0'] + k['size']] = rng.standard_normal((n, S.shape[1])) @ S.T
        return out

    def draw_prior(self, rng, n):
        """(n, P) deviates with the regulated prior covariance."""
        out = np.zeros((n, self.prior_mean.size))
        for idx, S in self.prior_S:
            if S.ndim == 1:
                out[:, idx] = rng.standard_normal((n, idx.size)) * S
            else:
                out[:, idx] = rng.standard_normal((n, S.shape[1])) @ S.T
        return out

    # -- device-facing packing ------------------------------------------------------
    def block_arrays(self, rows=None):
        """(row0, size, modes, tri, wt_flat) for the blocks inside rows [a, b)
        (row0 relative to a); wt_flat is a CUDA tensor when every block's weights were made on
        the device, a numpy array otherwise."""
        a, b = (0, self.n_data) if rows is None else rows
        sel = [k for k in self.blocks if k['row0'] >= a and k['row0'] + k['size'] <= b]
        for k in self.blocks:
            inside = k['row0'] >= a and k['row0'] + k['size'] <= b
            outside = k['row0'] + k['size'] <= a or k['row0'] >= b
            if not (inside or outside):
                raise ValueError('a shard boundary cuts through a covariance block')
        row0 = np.array([k['row0'] - a for k in sel], np.int64)
        size = np.array([k['size'] for k in sel], np.int64)
        modes = np.array([k['modes'] for k in sel], np.int64)
        tri = np.array([k['tri'] for k in sel], np.int32)
        if sel and all('Wt_dev' in k for k in sel):
            import torch                                  # weights made on the device stay there
            from . import _lib
            if len(sel) == 1:
                return row0, size, modes, tri, sel[0]['Wt_dev'].reshape(-1).contiguous()
            side = _lib.side_stream(sel[0]['Wt_dev'].device)
            with torch.cuda.stream(side):                 # (not the legacy default stream: _lib.side_stream)
                wt = torch.cat([k['Wt_dev'].reshape(-1) for k in sel]).contiguous()
            side.synchronize()
            return row0, size, modes, tri, wt
        wt = (np.concatenate([k['Wt'].reshape(-1) for k in sel]) if sel else np.zeros(0))
        return row0, size, modes, tri, np.ascontiguousarray(wt, np.float64)

    def prior_rows(self, p):
        """Whitened prior residual rows and their Jacobian, in the reference's order."""
        d = np.asarray(p, float) - self.prior_mean
        if self.prior_W[0] == 'diag':
            w = self.prior_W[1]
            return w * d, np.diag(w)
        _, diag_rows, Wrows = self.prior_W
        return diag_rows @ d, diag_rows, [W @ d for W in Wrows], Wrows


def _components(cov):
    """Connected components of the off-diagonal pattern of a symmetric matrix, each sorted, ordered
    by their first index (gvar.evalcov_blocks)."""
    n = cov.shape[0]
    seen = np.zeros(n, bool)
    adj = cov != 0.0
    out = []
    for i in range(n):
        if seen[i]:
            continue
        comp, stack = [], [i]
        seen[i] = True
        while stack:
            u = stack.pop()
            comp.append(u)
            for v in np.nonzero(adj[u] & ~seen)[0]:
                seen[v] = True
                stack.append(int(v))
        out.append(np.array(sorted(comp)))
    return out


def joint_whitening(ymean, yerr, prior_mean, prior_err, cross, svdcut=1e-12, engine=None, eps=None, noise=False, rng=None):
    """Whitening of concat(y, prior) when data and prior are correlated (``cross`` = the N x P
    covariance between them): what src/lsqfit/__init__.py:1892-1900 hands to gvar.PDF.

    The joint vector is permuted so that every covariance block (which may mix data and prior
    entries) is a contiguous row range, then regulated block by block like any data vector.  The
    result is a :class:`Whitening` WITHOUT a device-side prior: ``row_param[i] >= 0`` marks the rows
    that are prior entries (the device evaluates them as f_i = p_j), ``row_src[i]`` is the index of
    row i in concat(y, prior), ``model_rows`` the permuted positions of the data rows."""
    ymean = np.array(ymean, float).reshape(-1)
    pm = np.array(prior_mean, float).reshape(-1)
    N, P = ymean.size, pm.size
    cross = np.asarray(cross, float)
    if cross.shape != (N, P):
        raise ValueError('cross must be the %d x %d covariance between data and prior' % (N, P))

    def dense(err, n):
        err_a = None if isinstance(err, dict) else np.asarray(err, float)
        if err_a is not None and err_a.ndim == 2:
            return err_a.copy()
        sd, blocks, _ = _as_blocks(err, n)
        c = np.diag(sd ** 2)
        for r0, b in blocks:
            c[r0:r0 + b.shape[0], r0:r0 + b.shape[0]] = b
        return c
    full = np.zeros((N + P, N + P))
    full[:N, :N] = dense(yerr, N)
    full[N:, N:] = dense(prior_err, P)
    full[:N, N:] = cross
    full[N:, :N] = cross.T
    # noise (src/lsqfit/__init__.py:247-256): noise[1] moves the prior means by a draw from the prior's own covariance before
    # anything else (:535-536); noise[0] adds, to the JOINT vector, a draw from what the regulation added (:1896, gvar.PDF's noise)
    noise = (bool(noise), bool(noise)) if np.ndim(noise) == 0 else (bool(noise[0]), bool(noise[1]))
    gen = np.random.default_rng(rng)
    if noise[1]:
        w, v = np.linalg.eigh(full[N:, N:])
        pm = pm + v @ (np.sqrt(np.clip(w, 0.0, None)) * gen.standard_normal(P))
    comps = _components(full)
    perm = np.concatenate(comps)
    z = np.concatenate([ymean, pm])[perm]
    sd = np.sqrt(np.diag(full))[perm]
    blocks, r0 = [], 0
    for c in comps:
        if c.size > 1:
            blocks.append((r0, full[np.ix_(c, c)]))
        r0 += c.size
    wh = Whitening(z, dict(sdev=sd, blocks=blocks), svdcut=svdcut, engine=engine, eps=eps, noise=(noise[0], False),
                   rng=gen.integers(1 << 62))
    wh.noise = noise
    wh.joint = True
    wh.row_src = perm
    wh.row_param = np.where(perm >= N, perm - N, -1).astype(np.int32)
    wh.model_rows = np.nonzero(perm < N)[0]
    wh.n_model = N
    wh.prior_mean_host = pm                       # host-side bookkeeping only (default p0, maxit = 0)
    wh.prior_sdev = np.sqrt(np.diag(full))[N:]
    wh.prior_cov_host = full[N:, N:]
    wh.data_cov_host = full[:N, :N]               # (resample.py: simulated copies without prior noise lose the cross terms)
    return wh


def rows_whitening(ymean, yerr, prior_mean, prior_err, svdcut=1e-12, udata=False):
    """Whitening of concat(y, prior) with the prior entries as ROWS (``row_param``, as :func:`joint_whitening` -- but without a
    dense (N + P)^2 matrix) and every correlated block whitened in its eigen basis, rows = modes, as gvar.PDF does
    (src/lsqfit/__init__.py:1892-1900).  For fits that are not invariant under a rotation of a block's whitened rows: scipy's
    robust losses act on each element of the residual vector the plugin hands over (src/lsqfit/_scipy.py:147-153)."""
    ymean = np.array(ymean, float).reshape(-1)
    N = ymean.size
    ysd, yblocks, yperm = _as_blocks(yerr, N)
    if udata:
        if yperm is not None:
            ysd = ysd[np.argsort(yperm)]
        yblocks, yperm = [], None
    if prior_mean is None:
        wh = Whitening(ymean, yerr if not udata else ysd, svdcut=svdcut, engine='eig')
        return wh
    pm = np.array(prior_mean, float).reshape(-1)
    P = pm.size
    psd, pblocks, pperm = _as_blocks(prior_err, P)
    if yperm is not None or pperm is not None:
        # interleaved covariance components: the joint construction permutes data and prior entries alike (dense, small fits)
        # (udata: the correlations of y are dropped, src/lsqfit/__init__.py:1892-1893 -- ysd is back in the caller's order)
        wh = joint_whitening(ymean, ysd if udata else yerr, pm, prior_err, np.zeros((N, P)), svdcut=svdcut, engine='eig')
        wh.rows_only = True                           # prior entries travel as rows, but nothing correlates them with the data
        return wh
    z = np.concatenate([ymean, pm])
    wh = Whitening(z, dict(sdev=np.concatenate([ysd, psd]), blocks=yblocks + [(N + r0, c) for r0, c in pblocks]), svdcut=svdcut,
                   engine='eig')
    wh.joint = True
    wh.row_src = np.arange(N + P)
    wh.row_param = np.concatenate([np.full(N, -1), np.arange(P)]).astype(np.int32)
    wh.model_rows = np.arange(N)
    wh.n_model = N
    wh.prior_mean_host = pm
    wh.prior_sdev = psd
    pc = np.diag(psd ** 2)
    for r0, c in pblocks:
        pc[r0:r0 + c.shape[0], r0:r0 + c.shape[0]] = c
    wh.prior_cov_host = pc
    wh.data_cov_host = dict(sdev=ysd, blocks=yblocks)       # (resample.py: simulated copies without prior noise)
    wh.rows_only = True
    return wh
