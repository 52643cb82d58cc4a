"""numpy model of the reduced-space structured interior point that the HIP kernel implements.

Test helper / executable specification (not product code, not the oracle): it exists so that the
algebra of neptune_amd/csrc (null-space tables, base rows, scatter-reduced normal matrix) can be
checked on CPU against the full-space oracle before and independently of the GPU.
"""
import numpy as np

A_POS_INV = np.array([
    [-0.03203276669713047, -0.09273093424558249, 0.3420572455666699, 1.1023313949144335],
    [-0.05111494245568798, -0.046272612998418894, 0.5458234872124772, 1.0979806946005568],
    [-0.07454781852812224, 0.203951949894552, 0.796048050105448, 1.0745478185281223],
    [1.0, 1.0, 0.9999999999999996, 0.9999999999999993]])
A_VEL_INV = np.array([
    [-0.07735026918962577, 0.16666666666666635, 1.077350269189625],
    [-0.07735026918962577, 0.49999999999999967, 1.077350269189625],
    [1.0000000000000002, 1.0000000000000009, 1.0000000000000016]])


def tables(K, T, weight, relaxed):
    """Everything that depends only on (K, T, weight, mode)."""
    M4 = A_POS_INV * np.array([T ** 3, T ** 2, T, 1.0])[:, None]
    V3 = A_VEL_INV * (np.array([3.0, 2.0, 1.0]) * np.array([T * T, T, 1.0]))[:, None]
    # theta (4K) = Phi a + PhiU [b0,c0,d0]; state (b,c,d) at segment starts, plus the end state
    Phi = np.zeros((4 * K, K)); PhiU = np.zeros((4 * K, 3))
    bcd_a = np.zeros((3, K)); bcd_u = np.eye(3)  # rows: b,c,d as functions of (a, init)
    end = None
    for i in range(K + 1):
        if i < K:
            Phi[4 * i + 0, i] = 1.0
            Phi[4 * i + 1] = bcd_a[0]; PhiU[4 * i + 1] = bcd_u[0]
            Phi[4 * i + 2] = bcd_a[1]; PhiU[4 * i + 2] = bcd_u[1]
            Phi[4 * i + 3] = bcd_a[2]; PhiU[4 * i + 3] = bcd_u[2]
            ea = np.zeros(K); ea[i] = 1.0
            nb_a = bcd_a[0] + 3 * T * ea; nb_u = bcd_u[0].copy()
            nc_a = bcd_a[1] + 2 * T * bcd_a[0] + 3 * T * T * ea; nc_u = bcd_u[1] + 2 * T * bcd_u[0]
            nd_a = bcd_a[2] + T * bcd_a[1] + T * T * bcd_a[0] + T ** 3 * ea; nd_u = bcd_u[2] + T * bcd_u[1] + T * T * bcd_u[0]
            bcd_a = np.stack([nb_a, nc_a, nd_a]); bcd_u = np.stack([nb_u, nc_u, nd_u])
        else:
            end = (bcd_a.copy(), bcd_u.copy())
    (eb, ec, ed), (ub, uc, ud) = end  # b_K, c_K, d_K
    if relaxed:
        N = np.eye(K); Pp = np.zeros((K, 3))
    else:
        Et = np.stack([eb, ec]); Ft = np.stack([ub, uc])  # Et a + Ft init = 0
        Pp = -np.linalg.pinv(Et) @ Ft
        if K > 2:
            Qf, _ = np.linalg.qr(Et.T, mode="complete")
            N = Qf[:, 2:]
        else:
            N = np.zeros((K, 0))
    nz = N.shape[1]
    Th = Phi @ N; ThU = Phi @ Pp + PhiU
    R = 8 * K
    B = np.zeros((R, nz)); U = np.zeros((R, 3))
    for i in range(K):
        for k in range(4):
            B[4 * i + k] = M4[:, k] @ Th[4 * i:4 * i + 4]; U[4 * i + k] = M4[:, k] @ ThU[4 * i:4 * i + 4]
        for k in range(3):
            B[4 * K + 3 * i + k] = V3[:, k] @ Th[4 * i:4 * i + 3]; U[4 * K + 3 * i + k] = V3[:, k] @ ThU[4 * i:4 * i + 3]
        B[7 * K + i] = 6 * T * Th[4 * i] + 2 * Th[4 * i + 1]; U[7 * K + i] = 6 * T * ThU[4 * i] + 2 * ThU[4 * i + 1]
    ep = ed @ N; up = ed @ Pp + ud
    ev = ec @ N; uv = ec @ Pp + uc
    eacc = 2 * (eb @ N); uacc = 2 * (eb @ Pp + ub)
    Hax = 72 * T * (N.T @ N) + 2 * weight * np.outer(ep, ep)
    Gi = 72 * T * (N.T @ Pp) + 2 * weight * np.outer(ep, up)
    if relaxed:
        Hax += 2 * weight * (np.outer(ev, ev) + np.outer(eacc, eacc))
        Gi += 2 * weight * (np.outer(ev, uv) + np.outer(eacc, uacc))
    # equality residual map for K<=2 (primary): [b_K, c_K] at a = Pp init
    res_u = np.stack([eb @ Pp + ub, ec @ Pp + uc])
    return dict(K=K, T=T, nz=nz, N=N, Pp=Pp, Th=Th, ThU=ThU, B=B, U=U, ep=ep, up=up, ev=ev, uv=uv, ea=eacc, ua=uacc,
                Hax=Hax, Gi=Gi, res_u=res_u, relaxed=relaxed, weight=weight)


def full_cost(theta, T, weight, final, relaxed):
    K = theta.shape[1]
    tp = np.array([T ** 3, T ** 2, T, 1.0]); qv = np.array([3 * T * T, 2 * T, 1.0, 0]); qa = np.array([6 * T, 2.0, 0, 0])
    c = 36 * T * (theta[:, :, 0] ** 2).sum()
    for ax in range(3):
        c += weight * (tp @ theta[ax, K - 1] - final[ax]) ** 2
        if relaxed:
            c += weight * ((qv @ theta[ax, K - 1]) ** 2 + (qa @ theta[ax, K - 1]) ** 2)
    return c


def solve(tb, coeff_init, mins, maxs, v_max, a_max, line_seg, line_nd, maxit=100, verbose=False):
    """Returns (ok, theta[3][K][4], objective, iters)."""
    K, T, nz, w = tb["K"], tb["T"], tb["nz"], tb["weight"]
    B, U = tb["B"], tb["U"]
    R = 8 * K
    init = coeff_init[:, 0, 1:4]  # [ax][b0,c0,d0]
    tp = np.array([T ** 3, T ** 2, T, 1.0])
    final = coeff_init[:, K - 1, :] @ tp
    has_qc = np.linalg.norm(coeff_init[:, 0, 3] - final) < 1.0
    off = U @ init.T  # [R][3]
    # rows: (alpha_x, alpha_y, alpha_z, rho, h)
    rows = []
    for ax in range(3):
        for rho in range(R):
            hi = maxs[ax] if rho < 4 * K else (v_max if rho < 7 * K else a_max)
            lo = mins[ax] if rho < 4 * K else (-v_max if rho < 7 * K else -a_max)
            a = np.zeros(3); a[ax] = 1.0
            rows.append((a, rho, hi)); rows.append((-a, rho, -lo))
    for sgm, nd in zip(line_seg, line_nd):
        for k in range(4):
            rows.append((np.array([nd[0], nd[1], 0.0]), 4 * sgm + k, 1 - nd[2]))
    al = np.array([r[0] for r in rows]); rho = np.array([r[1] for r in rows]); h = np.array([r[2] for r in rows], dtype=float)
    m = len(rows)

    def theta_of(z):
        return np.stack([(tb["Th"] @ z[ax] + tb["ThU"] @ init[ax]).reshape(K, 4) for ax in range(3)])

    def rowvals(cp):  # cp [R][3]
        return (al * cp[rho]).sum(1)

    if not tb["relaxed"] and K <= 2:
        z = np.zeros((3, 0))
        th = theta_of(z)
        res = tb["res_u"] @ init.T
        a = rowvals(off)
        pend = tb["up"] @ init.T
        ok = np.abs(res).max() <= 1e-6 and (a - h).max() <= 1e-6 and (not has_qc or ((pend - final) ** 2).sum() <= 0.01 + 1e-6)
        return ok, th, full_cost(th, T, w, final, False), 0
    a_guess = coeff_init[:, :, 0]
    # orthogonal projection of the guess's coefficients onto {Th z + ThU init} (QpTable.Zp in the kernel)
    z = np.stack([np.linalg.lstsq(tb["Th"], coeff_init[ax, :K].reshape(-1) - tb["ThU"] @ init[ax], rcond=None)[0] for ax in range(3)])
    g = np.stack([tb["Gi"] @ init[ax] - 2 * w * tb["ep"] * final[ax] for ax in range(3)])
    obj0 = full_cost(theta_of(np.zeros((3, nz))), T, w, final, tb["relaxed"])
    Hax = tb["Hax"]
    cp = B @ z.T + off
    a = rowvals(cp)
    s = np.maximum(h - a, 0.1); lam = 2.0 / s
    sq = lq = 0.0
    if has_qc:
        pend = z @ tb["ep"] + init @ tb["up"]
        c = ((pend - final) ** 2).sum() - 0.01
        sq = max(-c, 1e-3); lq = 1.0 / sq
    mt = m + (1 if has_qc else 0)
    qscale = max(1.0, np.abs(g).max())
    loose = None; stall = 0; first_loose = None; best_merit = 0.0; ntrig = 0; give_up = False
    for it in range(maxit):
        cp = B @ z.T + off
        a = rowvals(cp)
        rp = a + s - h
        Tl = np.zeros((R, 3)); np.add.at(Tl, rho, al * lam[:, None])
        rd = (Hax @ z.T).T + g + (B.T @ Tl).T
        if has_qc:
            pend = z @ tb["ep"] + init @ tb["up"]
            c = ((pend - final) ** 2).sum() - 0.01
            gq = 2 * (pend - final)[:, None] * tb["ep"][None, :]
            rpq = c + sq
            rd = rd + lq * gq
        mu = (s @ lam + sq * lq) / mt
        nrp = max(np.abs(rp).max(), abs(rpq) if has_qc else 0.0); nrd = np.abs(rd).max()
        obj = obj0 + sum(0.5 * z[ax] @ Hax @ z[ax] + g[ax] @ z[ax] for ax in range(3))
        gap = mu * mt
        if verbose:
            print(it, "rp %.2e rd %.2e gap %.2e obj %.9g" % (nrp, nrd, gap, obj))
        if nrp <= 1e-9 and nrd <= 1e-9 * qscale and gap <= 1e-10 * (1 + abs(obj)):
            return True, theta_of(z), obj, it
        is_loose = nrp <= 1e-6 and nrd <= 1e-6 * qscale and gap <= 1e-7 * (1 + abs(obj))
        if is_loose or first_loose is not None:   # keep the loosely converged iterate closest to the strict test, stop 3 after the first
            merit = max(nrp * 1e9, nrd / qscale * 1e9, gap / (1 + abs(obj)) * 1e10)
            last = first_loose is not None and it - first_loose >= 3
            if first_loose is None:
                first_loose = it
            if is_loose and (loose is None or merit < best_merit):
                loose = (z.copy(), obj); best_merit = merit
            if last:
                break
        W = lam / s
        D = np.zeros((R, 3, 3)); np.add.at(D, rho, W[:, None, None] * al[:, :, None] * al[:, None, :])
        n = 3 * nz
        M = np.zeros((n, n))
        for a1 in range(3):
            for a2 in range(3):
                M[a1 * nz:(a1 + 1) * nz, a2 * nz:(a2 + 1) * nz] = B.T @ (D[:, a1, a2][:, None] * B)
            M[a1 * nz:(a1 + 1) * nz, a1 * nz:(a1 + 1) * nz] += Hax
        if has_qc:
            wq = lq / sq
            for a1 in range(3):
                M[a1 * nz:(a1 + 1) * nz, a1 * nz:(a1 + 1) * nz] += lq * 2 * np.outer(tb["ep"], tb["ep"])
            M += wq * np.outer(gq.reshape(-1), gq.reshape(-1))
        try:
            L = np.linalg.cholesky(M)
        except np.linalg.LinAlgError:
            break
        sigma = 0.0; dsa = dla = None; dsqa = dlqa = 0.0
        for pas in range(2):
            if pas == 0:
                rc = s * lam; rcq = sq * lq
            else:
                smu = max(sigma * mu, 0.1 * 1e-10 * (1 + abs(obj)) / mt)   # never aim below a tenth of the strict gap
                if it >= 10 and aaff < 0.1:      # short affine step: the predictor is discarded (see qp_solve in the oracle)
                    ntrig += 1
                    if ntrig > 8:
                        give_up = True
                        break
                    dsa, dla = -rp, -lam + W * rp
                    if has_qc:
                        dsqa, dlqa = -rpq, -lq + wq * rpq
                rc = s * lam - smu + dsa * dla; rcq = sq * lq - smu + dsqa * dlqa
            v = rc / s - W * rp
            T1 = np.zeros((R, 3)); np.add.at(T1, rho, al * v[:, None])
            rhs = -rd + (B.T @ T1).T
            if has_qc:
                rhs = rhs + gq * (rcq / sq - wq * rpq)
            dx = np.linalg.solve(L.T, np.linalg.solve(L, rhs.reshape(-1))).reshape(3, nz)
            u = B @ dx.T
            gdx = rowvals(u)
            ds = -rp - gdx; dl = -rc / s + W * (rp + gdx)
            alpha = 1.0
            neg = ds < 0
            if neg.any():
                alpha = min(alpha, (-s[neg] / ds[neg]).min())
            neg = dl < 0
            if neg.any():
                alpha = min(alpha, (-lam[neg] / dl[neg]).min())
            if has_qc:
                gdxq = (gq * dx).sum()
                dsq = -rpq - gdxq; dlq = -rcq / sq + wq * (rpq + gdxq)
                if dsq < 0:
                    alpha = min(alpha, -sq / dsq)
                if dlq < 0:
                    alpha = min(alpha, -lq / dlq)
            if pas == 0:
                mua = ((s + alpha * ds) @ (lam + alpha * dl) + ((sq + alpha * dsq) * (lq + alpha * dlq) if has_qc else 0.0)) / mt
                sigma = (mua / mu) ** 3
                aaff = alpha
                dsa, dla = ds, dl
                if has_qc:
                    dsqa, dlqa = dsq, dlq
        if give_up:
            break
        alpha = min(1.0, min(max(1.0 - mu, 0.999), 0.99999) * alpha)
        if alpha < 1e-8:
            stall += 1
            if stall >= 3:
                break
        else:
            stall = 0
        z = z + alpha * dx; s = s + alpha * ds; lam = lam + alpha * dl
        if has_qc:
            sq += alpha * dsq; lq += alpha * dlq
    if loose is not None:
        return True, theta_of(loose[0]), loose[1], it
    return False, None, None, maxit


def optimize(K, T, weight, coeff_init, mins, maxs, v_max, a_max, line_seg, line_nd):
    """PolySolverGurobi::optimize semantics (solver_gurobi_poly.cpp:804-887)."""
    tp = np.array([T ** 3, T ** 2, T, 1.0])
    final = coeff_init[:, K - 1, :] @ tp
    ok, th, obj, it = solve(tables(K, T, weight, False), coeff_init, mins, maxs, v_max, a_max, line_seg, line_nd)
    status = 0
    if not ok:
        ok, th, obj, it = solve(tables(K, T, weight, True), coeff_init, mins, maxs, v_max, a_max, line_seg, line_nd)
        status = 1
    if not ok:
        return 2, coeff_init.copy(), float("nan"), it
    th = th.copy()
    if np.hypot(coeff_init[0, 0, 3] - final[0], coeff_init[1, 0, 3] - final[1]) < 1.0:
        th[2] = coeff_init[2]
    return status, th, obj, it
