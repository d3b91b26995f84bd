"""CPU oracle for the sdflabel differentiable-SDF-renderer hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement of the reference's algorithm (dense N x P
formulation, the same operation order as the reference's ATen graph), each
function citing the reference file:line it follows (paths relative to
/root/reference).  It exists so that the HIP kernels have an independent checker
on the GPU box, where the Python reference cannot travel.

  * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import this module, and only as the checker / the timed CPU baseline.  The
    product package (sdflabel_amd/) never imports it and has no CPU fallback.
  * Parity pinning: the reference ships no tests or golden vectors for this path
    (SURVEY.md §4).  The oracle is pinned against golden vectors generated IN
    THE BUILD CONTAINER by importing the reference itself
    (tools/make_golden.py -> tests/golden/*.npz); tests/test_oracle_golden.py
    checks every function below against them.
  * The backward functions restate what torch autograd computes for the
    reference graph (detach points: primitives.py:226,228; in-place eps
    assignment primitives.py:210; clamp sub-gradients) and are pinned against
    autograd gradients captured from the reference (golden G7; r03: also in the
    reference pipeline's cropped, off-centre camera regime, golden G14).
  * sphere_trace / sphere_trace_backward (end of the file) are the oracle of the
    sphere-tracing render mode, which the reference does NOT have:
    PARITY UNPINNED for that mode -- there is no reference output; the functions
    define the mode's arithmetic and are checked for self-consistency only
    (tests/test_oracle_golden.py::test_sphere_trace_oracle_self_consistency).

All functions take/return numpy arrays; `dtype` follows the inputs (float32 for
parity, float64 for threshold-margin analysis).
"""
import numpy as np

# --------------------------------------------------------------------------------------
# Grid  (sdfrenderer/grid.py)
# --------------------------------------------------------------------------------------


def generate_point_grid(density):
    """grid.py:22-41 -- staggered D^3 sample grid, z fastest, odd flat indices shifted in x,y by 1/D."""
    c = density * 1j
    X, Y, Z = np.mgrid[-1:1:c, -1:1:c, -1:1:c]
    g = np.concatenate((X[..., None], Y[..., None], Z[..., None]), axis=-1).reshape((-1, 3))
    g[1::2, :2] += ((X.max() - X.min()) / density / 2)
    return g.astype(np.float32)


def get_surface_points(points, sdf, grad_points, threshold=0.03):
    """grid.py:43-71 -- zero-isosurface projection.

    points (G,3), sdf (G,1), grad_points (G,3) = d(sum sdf)/d points (what the
    autograd hook grid.py:11-12,20 captures).  Returns points_masked (N,3),
    nocs (N,3), normals_masked (N,3), band index (N,), n_hat (G,3).
    """
    dt = sdf.dtype
    nrm = np.sqrt(np.sum(grad_points * grad_points, axis=1)).astype(dt)          # grid.py:57
    with np.errstate(invalid="ignore", divide="ignore"):
        n_hat = (grad_points / nrm[:, None]).astype(dt)                           # grid.py:58
    proj = (points - sdf * n_hat).astype(dt)                                      # grid.py:61
    mask = (np.abs(sdf) < dt.type(threshold))[:, 0]                               # grid.py:64
    idx = np.nonzero(mask)[0]
    pm = proj[idx]                                                                # grid.py:65
    nm = n_hat[idx]                                                               # grid.py:66
    nocs = ((pm + dt.type(1)) / dt.type(2)).astype(dt)                            # grid.py:67
    return pm, nocs, nm, idx, n_hat


def get_surface_points_backward(sdf, n_hat, idx, g_points_masked, g_nocs=None):
    """Autograd of grid.py:61-67 w.r.t. sdf and grid points (n_hat is a constant: it comes from a
    .grad tensor, grid.py:56-58).  Returns g_sdf (G,1), g_gridpoints (G,3)."""
    dt = sdf.dtype
    g = g_points_masked.astype(dt).copy()
    if g_nocs is not None:
        g = g + g_nocs.astype(dt) / dt.type(2)
    g_sdf = np.zeros_like(sdf)
    g_pts = np.zeros((sdf.shape[0], 3), dt)
    g_pts[idx] = g
    g_sdf[idx, 0] = -np.sum(g * n_hat[idx], axis=1)
    return g_sdf, g_pts


# --------------------------------------------------------------------------------------
# DeepSDF decoder  (sdfrenderer/deepsdf/networks/deep_sdf_decoder_scale.py)
# --------------------------------------------------------------------------------------


def fold_weight_norm(v, g):
    """torch.nn.utils.weight_norm(dim=0) as used at deep_sdf_decoder_scale.py:51-52: w = v * (g / ||v||_row)."""
    nrm = np.sqrt(np.sum(v.astype(np.float32) ** 2, axis=1, keepdims=True)).astype(np.float32)
    return (v * (g.reshape(-1, 1) / nrm)).astype(np.float32)


def decoder_layers_from_state(state, spec):
    """Build the effective per-layer (W[out,in], b[out], ln) list from a reference state dict
    (keys lin{l}.weight | lin{l}.weight_g/_v | lin{l}.bias | bn{l}.weight/bias), float32."""
    n_lin = len(spec["dims"]) + 1
    layers = []
    for l in range(n_lin):
        p = "lin%d." % l
        if p + "weight_v" in state:
            W = fold_weight_norm(np.asarray(state[p + "weight_v"], np.float32), np.asarray(state[p + "weight_g"], np.float32))
        else:
            W = np.asarray(state[p + "weight"], np.float32)
        b = np.asarray(state[p + "bias"], np.float32)
        ln = None
        if ("bn%d.weight" % l) in state:
            ln = (np.asarray(state["bn%d.weight" % l], np.float32), np.asarray(state["bn%d.bias" % l], np.float32))
        layers.append((W, b, ln))
    return layers


def decoder_forward(layers, spec, inputs, want_cache=False):
    """deep_sdf_decoder_scale.py:78-107 (eval mode: dropout inactive :103-104).

    inputs (G, L+3) = [latent, xyz].  Returns sdf (G,1) (and the per-layer cache for the backward).
    spec keys used: latent_in, xyz_in_all, use_tanh (norm handled through `ln` in layers).
    """
    dt = inputs.dtype
    latent_in = tuple(spec.get("latent_in", ()))
    xyz_in_all = bool(spec.get("xyz_in_all", False))
    use_tanh = bool(spec.get("use_tanh", False))
    n_lin = len(layers)
    xyz = inputs[:, -3:]
    x = inputs
    cache = []
    for l in range(n_lin):
        W, b, ln = layers[l]
        if l in latent_in:
            x = np.concatenate([x, inputs], axis=1)                                # :90-91
        elif l != 0 and xyz_in_all:
            x = np.concatenate([x, xyz], axis=1)                                   # :92-93
        xin = x
        x = (x @ W.astype(dt).T + b.astype(dt)).astype(dt)                         # :94
        pre_tanh = None
        if l == n_lin - 1 and use_tanh:
            pre_tanh = x
            x = np.tanh(x)                                                         # :96-97
        lnc = None
        pre_relu = None
        if l < n_lin - 1:
            if ln is not None:                                                     # :99-101 LayerNorm variant
                mu = x.mean(axis=1, keepdims=True)
                var = ((x - mu) ** 2).mean(axis=1, keepdims=True)
                rstd = 1.0 / np.sqrt(var + dt.type(1e-5))
                xh = (x - mu) * rstd
                lnc = (xh, rstd)
                x = (xh * ln[0].astype(dt) + ln[1].astype(dt)).astype(dt)
            pre_relu = x
            x = np.maximum(x, dt.type(0))                                          # :102
        if want_cache:
            cache.append((xin.shape[1], pre_relu, lnc, pre_tanh))
    out_pre = x
    x = np.tanh(x).astype(dt)                                                      # :106-107 (self.th always present)
    if want_cache:
        return x, (cache, out_pre)
    return x


def decoder_backward_inputs(layers, spec, inputs, cache, g_out):
    """Autograd of decoder_forward w.r.t. `inputs` only (weights frozen: optimizer.py:34-38 optimises
    yaw/trans/scale/latent).  g_out (G,1) -> g_inputs (G, L+3).  With g_out = 1 this is what the
    grid hook captures for the xyz columns (grid.py:55-56)."""
    dt = inputs.dtype
    latent_in = tuple(spec.get("latent_in", ()))
    xyz_in_all = bool(spec.get("xyz_in_all", False))
    n_lin = len(layers)
    caches, out_pre = cache
    th = np.tanh(out_pre)
    g = (g_out * (dt.type(1) - th * th)).astype(dt)                                # d tanh
    g_inputs = np.zeros_like(inputs)
    n_in0 = inputs.shape[1]
    for l in range(n_lin - 1, -1, -1):
        W, b, ln = layers[l]
        in_dim, pre_relu, lnc, pre_tanh = caches[l]
        if l < n_lin - 1:
            g = g * (pre_relu > 0)                                                 # relu'
            if ln is not None:
                xh, rstd = lnc
                gy = g * ln[0].astype(dt)
                m1 = gy.mean(axis=1, keepdims=True)
                m2 = (gy * xh).mean(axis=1, keepdims=True)
                g = (gy - m1 - xh * m2) * rstd
        elif pre_tanh is not None:
            t = np.tanh(pre_tanh)
            g = g * (dt.type(1) - t * t)
        g = (g @ W.astype(dt)).astype(dt)                                          # (G, in_dim)
        if l in latent_in:
            g_inputs += g[:, -n_in0:]
            g = g[:, :-n_in0]
        elif l != 0 and xyz_in_all:
            g_inputs[:, -3:] += g[:, -3:]
            g = g[:, :-3]
    g_inputs += g
    return g_inputs


def scale_net_forward(scale_params, latent_row):
    """deep_sdf_decoder_scale.py:69-75,112 -- 3 tiny linears with ReLU on the first row's latent."""
    x = latent_row
    for i, (W, b) in enumerate(scale_params):
        x = x @ W.T + b
        if i < len(scale_params) - 1:
            x = np.maximum(x, 0)
    return x


# --------------------------------------------------------------------------------------
# Pose helpers  (utils/refinement.py, pipelines/optimizer.py, renderer/utils_rasterer.py)
# --------------------------------------------------------------------------------------


def rot_from_yaw(yaw, dtype=np.float32):
    """utils/refinement.py:108-125."""
    c, s = np.cos(dtype(yaw)), np.sin(dtype(yaw))
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype)


def render_pose(yaw, trans, dtype=np.float32):
    """pipelines/optimizer.py:87-90 -- [R_y(yaw)|t], row 1 negated BEFORE the translation is written."""
    P = np.eye(4, dtype=dtype)
    P[:3, :3] = rot_from_yaw(yaw, dtype)
    P[1] *= -1
    P[:3, 3] = np.asarray(trans, dtype)
    return P


def calibration_matrix(resolution_px, diagonal_mm, focal_len_mm, skew=0.0):
    """renderer/utils_rasterer.py:59-83."""
    rx, ry = resolution_px
    diag = np.sqrt(rx ** 2 + ry ** 2)
    rx_mm = rx / diag * diagonal_mm
    ry_mm = ry / diag * diagonal_mm
    ax = focal_len_mm * (rx / rx_mm)
    ay = focal_len_mm * (ry / ry_mm)
    return np.array([[ax, skew, rx / 2], [0, ay, ry / 2], [0, 0, 1]])


def qrot(q, v):
    """renderer/utils_rasterer.py:6-24."""
    qvec = q[:, 1:]
    uv = np.cross(qvec, v)
    uuv = np.cross(qvec, uv)
    return v + 2 * (q[:, :1] * uv + uuv)


def pixel_grid(resolution_px):
    """renderer/rasterer.py:25-27 -- (P,2) integer pixel coordinates, x fastest."""
    rx, ry = resolution_px
    yy, xx = np.mgrid[0:ry, 0:rx]
    return np.concatenate((xx[..., None], yy[..., None]), axis=-1).reshape((-1, 2))


# --------------------------------------------------------------------------------------
# Projection  (sdfrenderer/renderer/projection.py)
# --------------------------------------------------------------------------------------


def project_in_2D(K, camera_pose, points, normals, colors, resolution_px, output_nocs=True):
    """projection.py:7-101 (rot='dcm', filter_normals=True, filter_hpr=False)."""
    dt = K.dtype
    eps = np.finfo(dt).eps
    rx, ry = resolution_px
    RT = camera_pose[:-1, :].astype(dt)                                            # :34
    ones = np.ones((points.shape[0], 1), dt)
    ch = np.concatenate([points.astype(dt), ones], axis=-1).T                     # :44-46
    normals_p = (RT[:, :3] @ normals.astype(dt).T).T                              # :49
    if output_nocs:
        colors = points.astype(dt).copy()                                          # :53-55
        colors[:, 0] *= -1
    p3 = (RT @ ch).T                                                               # :58
    dot = np.sum(normals_p * p3, axis=1)                                           # :62 (bmm)
    keep = dot < 0
    fidx = np.nonzero(keep)[0]
    out = {
        "points_3d_filt": p3[fidx], "normals_3d_filt": normals_p[fidx], "colors_3d_filt": colors[fidx],
        "filt_idx": fidx, "dot": dot,
    }
    p2h = (K @ p3.T).T                                                             # :88
    p2 = p2h[:, :2] / (p2h[:, 2:] + eps)                                           # :89
    out["points_3d"] = p3
    out["normals_3d"] = normals_p
    out["colors_3d"] = colors
    out["points_2d"] = np.concatenate([np.clip(p2[:, 0:1], -1, rx), np.clip(p2[:, 1:2], -1, ry)], axis=-1)  # :92-93,:99
    return out


def project_in_2D_quat(K, camera_pose, points, normals, colors, resolution_px, output_nocs=True):
    """projection.py:104-199 (filter_normals=False default: no *_filt keys; NOCS x not flipped :147-149)."""
    dt = K.dtype
    eps = np.finfo(dt).eps
    rx, ry = resolution_px
    q = camera_pose[:4].astype(dt)
    t = camera_pose[4:].astype(dt)
    qn = np.broadcast_to(q[None], (normals.shape[0], 4))
    normals_p = qrot(qn, normals.astype(dt))                                       # :136-137
    if output_nocs:
        colors = points.astype(dt).copy()
    p3 = qrot(qn, points.astype(dt)) + t[None]                                     # :152-157
    p2h = (K @ p3.T).T
    p2 = p2h[:, :2] / (p2h[:, 2:] + eps)
    return {
        "points_3d": p3, "normals_3d": normals_p, "colors_3d": colors,
        "points_2d": np.concatenate([np.clip(p2[:, 0:1], -1, rx), np.clip(p2[:, 1:2], -1, ry)], axis=-1),
    }


# --------------------------------------------------------------------------------------
# Primitive: 3-D tangent disc  (sdfrenderer/renderer/primitives.py:165-242)
# --------------------------------------------------------------------------------------


def pixel_rays(Kinv, grid_2d):
    """primitives.py:203-208 -- rays K^-1 [x,y,1] for every pixel, (P,3)."""
    dt = Kinv.dtype
    g = np.concatenate([grid_2d.astype(dt), np.ones((grid_2d.shape[0], 1), dt)], axis=-1)
    return (Kinv @ g.T).T.astype(dt)


def inside_surfel(Kinv, grid_2d, vertex_3d, normals, diam=0.04, depth_constant=150, add_bg=False,
                  chunk=8192, want_aux=False, softclamp=False, softclamp_constant=5):
    """primitives.py:165-242; softclamp=False is the renderer's call (rasterer.py:102-104), softclamp=True the function's own default
    (:174,:217-218: the coverage mask is sigmoid((diam - d) * c) > 0, true until exp overflows -- d < diam + 88.7 / c).

    Kinv is K.float().inverse() (primitives.py:204), supplied by the caller.  Returns the (N[+1], P)
    weight matrix (the reference expands it to 3 identical channels, :241).  Dense, chunked over pixels.
    """
    dt = vertex_3d.dtype
    eps = np.finfo(dt).eps
    fmin = np.finfo(dt).min
    N = vertex_3d.shape[0]
    P = grid_2d.shape[0]
    rays = pixel_rays(Kinv.astype(dt), grid_2d)
    n_v3d = np.sum(normals * vertex_3d, axis=1).astype(dt)                         # :202
    rows = N + (1 if add_bg else 0)
    W = np.zeros((rows, P), dt)
    aux = {"t": None, "mask": None, "margin_disc": np.full(P, np.inf), "margin_b": np.full(P, np.inf)} if want_aux else None
    for s in range(0, P, chunk):
        r = rays[s:s + chunk]                                                      # (p,3)
        b = (normals @ r.T).astype(dt)                                             # :209 (N,p)
        if want_aux:
            aux["margin_b"][s:s + chunk] = np.min(np.abs(np.abs(b) - 0.01), axis=0) if N else np.inf
        small = np.abs(b) < dt.type(0.01)
        b = np.where(small, dt.type(eps), b)                                       # :210
        z = (n_v3d[:, None] / b).astype(dt)                                        # :211
        g3 = r[None, :, :] * z[:, :, None]                                         # :212
        vec = vertex_3d[:, None, :] - g3                                           # :215
        d = np.sqrt(np.sum(vec * vec, axis=-1)).astype(dt)
        if softclamp:
            m = _sigmoid(((dt.type(diam) - d) * dt.type(softclamp_constant)).astype(dt)) > 0   # :217-218,:226
        else:
            dist = np.maximum(dt.type(diam) - d, dt.type(0))                       # :220
            m = (dist > 0)
        if want_aux:
            aux["margin_disc"][s:s + chunk] = np.min(np.abs(dt.type(diam) - d), axis=0) if N else np.inf
        mf = m.astype(dt)
        zz = (-z * mf).astype(dt)                                                  # :227
        zn = np.sqrt(np.sum(zz * zz, axis=0)).astype(dt)                           # :228
        zz = (np.maximum(zz / (zn[None] + dt.type(eps)) + dt.type(1), dt.type(0)) * dt.type(depth_constant)).astype(dt)  # :229-230
        if add_bg:
            z2d = -vertex_3d[:, 2] * dt.type(depth_constant)                       # :234
            zbg = np.full((1, zz.shape[1]), z2d.min() - 1, dt)                     # :235
            zz = np.concatenate([zz, zbg], axis=0)
            mf = np.concatenate([mf, np.ones((1, mf.shape[1]), dt)], axis=0)
            m = np.concatenate([m, np.ones((1, m.shape[1]), bool)], axis=0)
        zm = np.where(m, zz, dt.type(fmin))                                        # :240
        zm = zm - zm.max(axis=0, keepdims=True)
        e = np.exp(zm)
        W[:, s:s + chunk] = (e / e.sum(axis=0, keepdims=True) * mf).astype(dt)
    if want_aux:
        return W, aux
    return W


# --------------------------------------------------------------------------------------
# Secondary primitives: 2-D circle and 15x15 stamp  (sdfrenderer/renderer/primitives.py:4-162)
# --------------------------------------------------------------------------------------


def _sigmoid(x):
    with np.errstate(over="ignore"):
        return (1 / (1 + np.exp(-x))).astype(x.dtype)


def depth_logits(vertex_3d, depth_constant):
    """primitives.py:57-61 / :141-144 -- per-vertex logit clamp(-z/(||z||+eps) + 1, 0) * C  (the norm is detached).
    Returns (logits (N,), pre-clamp value q (N,), ||z||)."""
    dt = vertex_3d.dtype
    eps = np.finfo(dt).eps
    z = -vertex_3d[:, 2]
    zn = np.sqrt(np.sum(z * z)).astype(dt)
    q = (z / (zn + dt.type(eps)) + dt.type(1)).astype(dt)
    return (np.maximum(q, dt.type(0)) * dt.type(depth_constant)).astype(dt), q, zn


def inside_circle(K, grid_2d, vertex_2d, vertex_3d, diam=0.02, depth_constant=100, softclamp_constant=3, add_bg=False, want_mask=False,
                  softclamp=True):
    """primitives.py:4-71 with softclamp=True (the renderer's call, rasterer.py:94-96, leaves the default).  Note the reference
    thresholds sigmoid(.) > 0 (:55), which only fails once exp overflows (~29.6 px beyond the circle), and that the softmax runs
    over z * mask (:70): uncovered vertices keep a logit of 0 in the denominator."""
    dt = vertex_3d.dtype
    eps = np.finfo(dt).eps
    diff = vertex_2d[:, None, :2].astype(dt) - grid_2d[None].astype(dt)                    # :42
    r = np.abs(K[0, 0].astype(dt) * dt.type(diam) / (vertex_3d[:, 2] + dt.type(eps)))      # :47
    d = np.sqrt(np.sum(diff * diff, axis=-1)).astype(dt)
    if softclamp:
        m = _sigmoid(((r[:, None] - d) * dt.type(softclamp_constant)).astype(dt)) > 0      # :46-49,:55
    else:
        m = np.maximum(r[:, None] - d, dt.type(0)) > 0                                      # :51-53,:55: a hard circle
    zl, _, _ = depth_logits(vertex_3d, depth_constant)                                      # :56-61
    L = zl[:, None] * m.astype(dt)
    mf = m.astype(dt)
    if add_bg:                                                                              # :64-67
        L = np.concatenate([L, np.full((1, L.shape[1]), zl.min() - 1, dt)], axis=0)
        mf = np.concatenate([mf, np.ones((1, mf.shape[1]), dt)], axis=0)
    L = L - L.max(axis=0, keepdims=True)
    e = np.exp(L)
    W = (e / e.sum(axis=0, keepdims=True) * mf).astype(dt)                                  # :70
    return (W, m) if want_mask else W


def inside_circle_opt(K, vertex_2d, vertex_3d, diam=0.025, depth_constant=10000, softclamp_constant=5, add_bg=False, want_mask=False,
                      softclamp=True):
    """primitives.py:74-162 -- every vertex stamps a 15x15 pixel square (sigmoid weights, all > 0) at trunc(vertex_2d + offset),
    indices clamped into the image (:125-127), duplicates summed by the sparse tensor (:135-138); image size from K (:109-110)."""
    dt = vertex_3d.dtype
    eps = np.finfo(dt).eps
    fmin = np.finfo(dt).min
    x_px, y_px = int(K[0, 2]) * 2, int(K[1, 2]) * 2
    yy, xx = np.mgrid[-7:8, -7:8]
    off = np.concatenate((xx[..., None], yy[..., None]), axis=-1).reshape((-1, 2)).astype(dt)      # rasterer.py:30-32
    dist_prim = np.sqrt(np.sum(off * off, axis=-1)).astype(dt)
    r = np.abs(K[0, 0].astype(dt) * dt.type(diam) / (vertex_3d[:, 2] + dt.type(eps)))
    if softclamp:
        prim = _sigmoid(((r[:, None] - dist_prim[None]) * dt.type(softclamp_constant)).astype(dt))   # :118
    else:
        prim = np.maximum(r[:, None] - dist_prim[None], dt.type(0)).astype(dt)                       # :120: stamp offsets inside the circle only
    ids = np.trunc(off[None] + vertex_2d[:, None, :].astype(dt)).astype(np.int64)                   # :122-124
    ids[..., 0] = np.clip(ids[..., 0], 0, x_px - 1)
    ids[..., 1] = np.clip(ids[..., 1], 0, y_px - 1)
    N = vertex_3d.shape[0]
    dense = np.zeros((N, y_px, x_px), np.float32)
    np.add.at(dense, (np.arange(N)[:, None], ids[..., 1], ids[..., 0]), prim.astype(np.float32))
    m = dense.reshape(N, -1) > 0                                                                     # :155
    zl, _, _ = depth_logits(vertex_3d, depth_constant)
    L = np.broadcast_to(zl[:, None], m.shape).astype(dt)
    mf = m.astype(dt)
    if add_bg:
        L = np.concatenate([L, np.full((1, L.shape[1]), zl.min() - 1, dt)], axis=0)
        mf = np.concatenate([mf, np.ones((1, mf.shape[1]), dt)], axis=0)
        m2 = np.concatenate([m, np.ones((1, m.shape[1]), bool)], axis=0)
    else:
        m2 = m
    L = np.where(m2, L, dt.type(fmin))                                                               # :156
    L = L - L.max(axis=0, keepdims=True)
    e = np.exp(L)
    W = (e / e.sum(axis=0, keepdims=True) * mf).astype(dt)
    return (W, m) if want_mask else W


def circle_backward(W, m, vertex_3d, c_attr, n_attr, g_color, g_mask, g_depth, g_normals, depth_constant, unmasked_softmax, bg=None):
    """Autograd of inside_circle / inside_circle_opt + compositing w.r.t. the vertex depths (through the logits; masks, the depth
    norm and the 2-D positions carry no gradient) and the composited attributes.  W is the forward weight matrix (N[+1], P),
    m the (N, P) coverage.  unmasked_softmax: inside_circle's softmax over z*mask (uncovered logit 0).  Returns g_v3 (N,3),
    g_cattr (N,3), g_nattr (N,3)."""
    dt = vertex_3d.dtype
    eps = np.finfo(dt).eps
    N, P = m.shape
    zl, q, zn = depth_logits(vertex_3d, depth_constant)
    has_bg = W.shape[0] == N + 1
    mf = m.astype(np.float64)
    w = W[:N].astype(np.float64)
    z3 = vertex_3d[:, 2].astype(np.float64)
    zero3 = np.zeros((3, P))
    gC = g_color.reshape(3, P).astype(np.float64) if g_color is not None else zero3
    gM = g_mask.reshape(P).astype(np.float64) if g_mask is not None else np.zeros(P)
    gD = g_depth.reshape(P).astype(np.float64) if g_depth is not None else np.zeros(P)
    gN = g_normals.reshape(3, P).astype(np.float64) if g_normals is not None else zero3
    Cs = c_attr.T.astype(np.float64) @ w
    Ms = W.astype(np.float64).sum(axis=0)
    Ns = n_attr.T.astype(np.float64) @ w
    if has_bg:
        Cs = Cs + W[N][None].astype(np.float64) * bg.reshape(3, P)
    gCg = gC * (Cs.astype(dt) <= 1)
    gMg = gM * (Ms.astype(dt) <= 1)
    gNg = gN * (Ns.astype(dt) <= 1)
    dW = c_attr.astype(np.float64) @ gCg + gMg[None] + z3[:, None] * gD[None] + n_attr.astype(np.float64) @ gNg
    g_c = w @ gCg.T
    g_na = w @ gNg.T
    g_v3 = np.zeros((N, 3))
    g_v3[:, 2] += w @ gD
    # softmax backward.  sm == w on covered entries; uncovered entries have dL/dsm = 0 (w = sm * m)
    S = np.sum(w * dW, axis=0)
    if has_bg:
        dWbg = np.sum(gCg * bg.reshape(3, P), axis=0) + gMg
        S = S + W[N].astype(np.float64) * dWbg
    dlog = w * (dW - S[None])                      # covered rows only matter: logit = zl * m (or masked_fill)
    dzl = np.sum(dlog * mf, axis=1)
    if has_bg:                                     # z_bg = zl.min() - 1  (:65 / :147)
        dlog_bg = W[N].astype(np.float64) * (dWbg - S)
        dzl[int(np.argmin(zl))] += dlog_bg.sum()
    if not unmasked_softmax:
        pass                                       # masked_fill: identical on covered entries
    dq = dzl * depth_constant * (q >= 0)
    g_v3[:, 2] += dq * (-1.0 / (float(zn) + eps))  # q = -z/(zn+eps) + 1, zn detached
    return g_v3.astype(dt), g_c.astype(dt), g_na.astype(dt)


# --------------------------------------------------------------------------------------
# Rasterer  (sdfrenderer/renderer/rasterer.py:49-155)
# --------------------------------------------------------------------------------------


def rasterer_forward(K, Kinv, resolution_px, coords, normals, colors, camera_matrix, rot="dcm", bg=None,
                     output_mask=True, output_depth=True, output_normals=True, output_nocs=True, chunk=8192,
                     want_aux=False, primitives="disc"):
    """rasterer.py:49-155.  Returns (rendering dict, points dict, proj dict)."""
    dt = K.dtype
    rx, ry = resolution_px
    if rot == "dcm":
        proj = project_in_2D(K, camera_matrix, coords, normals, colors, resolution_px, output_nocs)
    else:
        proj = project_in_2D_quat(K, camera_matrix, coords, normals, colors, resolution_px, output_nocs)
    v3 = proj["points_3d"].astype(dt)
    nrm = proj["normals_3d"].astype(dt)
    col = proj["colors_3d"].astype(dt)
    grid_2d = pixel_grid(resolution_px)
    if primitives == "disc":                                                       # :101-104
        res = inside_surfel(Kinv, grid_2d, v3, nrm, diam=0.04, add_bg=(bg is not None), chunk=chunk, want_aux=want_aux)
        W, aux = res if want_aux else (res, None)
    elif primitives == "circle":                                                   # :93-96
        W, aux = inside_circle(K, grid_2d, proj["points_2d"], v3, diam=0.02, add_bg=(bg is not None)), None
    elif primitives == "circle_opt":                                               # :97-100
        W, aux = inside_circle_opt(K, proj["points_2d"], v3, diam=0.025, add_bg=(bg is not None)), None
    else:
        raise ValueError(primitives)
    rendering = {}
    if bg is not None:                                                             # :107-111
        color = (W[:-1].T @ ((col + 1) / 2)).T + W[-1][None, :] * bg.reshape(3, -1).astype(dt)
    else:
        c_t = (col + 1) / 2 if output_nocs else col                                # :113-116
        color = (W.T @ c_t).T
    rendering["color"] = np.minimum(color, 1).reshape(3, ry, rx).astype(dt)        # :123-124
    rendering["color_pre"] = color.reshape(3, ry, rx).astype(dt)
    if output_mask:
        msum = W.sum(axis=0)
        rendering["mask"] = np.minimum(msum, 1).reshape(1, ry, rx).astype(dt)      # :127-131
        rendering["mask_pre"] = msum.reshape(1, ry, rx).astype(dt)
    if output_depth and bg is None:
        rendering["depth"] = (W.T @ v3[:, 2]).reshape(1, ry, rx).astype(dt)        # :134-137
    if output_normals and bg is None:
        nsum = (W.T @ ((nrm + 1) / 2)).T
        rendering["normals"] = np.minimum(nsum, 1).reshape(3, ry, rx).astype(dt)   # :140-144
        rendering["normals_pre"] = nsum.reshape(3, ry, rx).astype(dt)
    points = {"xyz": v3, "rgb": (col + 1) / 2}                                     # :147-153
    if "points_3d_filt" in proj:
        points["xyzf"] = proj["points_3d_filt"]
        points["rgbf"] = (proj["colors_3d_filt"] + 1) / 2
    if want_aux:
        return rendering, points, proj, aux
    return rendering, points, proj


def splat_backward(Kinv, resolution_px, v3, nrm, c_attr, g_color, g_mask, g_depth, g_normals,
                   diam=0.04, depth_constant=150, chunk=4096, grid_2d=None):
    """Autograd of inside_surfel + compositing (primitives.py:202-241, rasterer.py:119-144; bg=None) w.r.t.
    camera-frame surfel positions v3 (N,3), camera-frame normals nrm (N,3) and the composited colour
    attribute c_attr (N,3) (= (colors+1)/2 for NOCS, `colors` otherwise).

    Detached in the reference and therefore constants here: the disc mask (:226), the per-pixel norm (:228);
    entries with |n.ray|<0.01 are overwritten in place by eps (:210) and pass no gradient to n through b.
    clamp(max=1)/clamp(min=0) pass gradient on the closed side (x<=1 / x>=0), as ATen does.
    g_* are gradients of the post-clamp images, shapes (3,H,W),(1,H,W),(1,H,W),(3,H,W) or None.
    Returns g_v3, g_nrm, g_cattr.
    """
    dt = v3.dtype
    eps = np.finfo(dt).eps
    fmin = np.finfo(dt).min
    N = v3.shape[0]
    if grid_2d is None:                       # grid_2d: optional subset of pixels (g_* then have P = len(grid_2d) columns)
        grid_2d = pixel_grid(resolution_px)
    P = grid_2d.shape[0]
    rays = pixel_rays(Kinv.astype(dt), grid_2d)
    a = np.sum(nrm * v3, axis=1).astype(dt)
    n_attr = ((nrm + 1) / 2).astype(dt)
    zero3 = np.zeros((3, P), dt)
    gC = g_color.reshape(3, P).astype(dt) if g_color is not None else zero3
    gM = g_mask.reshape(P).astype(dt) if g_mask is not None else np.zeros(P, dt)
    gD = g_depth.reshape(P).astype(dt) if g_depth is not None else np.zeros(P, dt)
    gN = g_normals.reshape(3, P).astype(dt) if g_normals is not None else zero3
    g_v3 = np.zeros((N, 3), np.float64)
    g_n = np.zeros((N, 3), np.float64)
    g_c = np.zeros((N, 3), np.float64)
    g_a = np.zeros(N, np.float64)
    for s in range(0, P, chunk):
        r = rays[s:s + chunk]
        b0 = (nrm @ r.T).astype(dt)
        small = np.abs(b0) < dt.type(0.01)
        b = np.where(small, dt.type(eps), b0)
        t = (a[:, None] / b).astype(dt)
        g3 = r[None] * t[:, :, None]
        vec = v3[:, None, :] - g3
        d = np.sqrt(np.sum(vec * vec, axis=-1)).astype(dt)
        m = (np.maximum(dt.type(diam) - d, dt.type(0)) > 0)
        mf = m.astype(dt)
        zz = (-t * mf).astype(dt)
        zn = np.sqrt(np.sum(zz * zz, axis=0)).astype(dt)
        q = (zz / (zn[None] + dt.type(eps)) + dt.type(1)).astype(dt)
        logit = (np.maximum(q, dt.type(0)) * dt.type(depth_constant)).astype(dt)
        zm = np.where(m, logit, dt.type(fmin))
        zm = zm - zm.max(axis=0, keepdims=True)
        e = np.exp(zm)
        sm = (e / e.sum(axis=0, keepdims=True)).astype(dt)
        w = sm * mf
        # composites (pre-clamp) and clamp gates
        Cs = c_attr.T @ w
        Ms = w.sum(axis=0)
        Ns = n_attr.T @ w
        gCg = gC[:, s:s + chunk] * (Cs <= 1)
        gMg = gM[s:s + chunk] * (Ms <= 1)
        gDg = gD[s:s + chunk]
        gNg = gN[:, s:s + chunk] * (Ns <= 1)
        # dL/dw (N,p)
        dW = c_attr @ gCg + gMg[None] + v3[:, 2:3] * gDg[None] + n_attr @ gNg
        # attribute grads
        g_c += w @ gCg.T
        g_n += (w @ gNg.T) * 0.5
        g_v3[:, 2] += w @ gDg
        # softmax backward (w = sm * mf ; masked entries have sm == 0 unless the whole column is masked)
        dsm = dW * mf
        dlog = sm * (dsm - np.sum(sm * dsm, axis=0, keepdims=True))
        dlog = dlog * m                                    # masked_fill blocks the gradient
        dq = dlog * dt.type(depth_constant) * (q >= 0)
        dzz = dq / (zn[None] + dt.type(eps))
        dt_ = -dzz * mf                                    # zz = -t * mask
        da = dt_ / b
        db = -dt_ * t / b
        db = np.where(small, 0, db)                        # in-place eps assignment blocks grad (:210)
        g_a += da.sum(axis=1)
        g_n += db @ r
    g_n += g_a[:, None] * v3
    g_v3 += g_a[:, None] * nrm
    return g_v3.astype(dt), g_n.astype(dt), g_c.astype(dt)


def project_backward_dcm(camera_pose, points, normals, g_p3, g_nrm, g_col, output_nocs=True,
                         filt_idx=None, g_p3_filt=None, g_col_filt=None):
    """Autograd of projection.py:34-70 (rot='dcm') w.r.t. points, normals, colors and the (4,4) pose.
    g_col is the gradient w.r.t. colors_3d (pre '(c+1)/2').  Gradients arriving through the filtered
    outputs (points_3d_filt / colors_3d_filt) are scattered back through filt_idx."""
    dt = points.dtype
    R = camera_pose[:3, :3].astype(dt)
    g_p3 = g_p3.astype(np.float64).copy()
    g_col = g_col.astype(np.float64).copy()
    if filt_idx is not None and g_p3_filt is not None:
        np.add.at(g_p3, filt_idx, g_p3_filt)
    if filt_idx is not None and g_col_filt is not None:
        np.add.at(g_col, filt_idx, g_col_filt)
    g_points = g_p3 @ R
    g_normals = g_nrm.astype(np.float64) @ R
    g_colors_in = None
    if output_nocs:
        g_points = g_points + g_col * np.array([-1.0, 1.0, 1.0])
    else:
        g_colors_in = g_col.astype(dt)
    g_pose = np.zeros((4, 4), np.float64)
    g_pose[:3, :3] = g_p3.T @ points + g_nrm.astype(np.float64).T @ normals
    g_pose[:3, 3] = g_p3.sum(axis=0)
    return g_points.astype(dt), g_normals.astype(dt), g_colors_in, g_pose.astype(dt)


# --------------------------------------------------------------------------------------
# Losses of the refinement loop  (pipelines/optimizer.py:166-237)
# --------------------------------------------------------------------------------------

def loss_3d(pcd_est, pcd_lidar, scale, threshold=0.2, want_grad=False):
    """optimizer.py:166-198 with pcd_frustum = pcd_lidar / scale (optimizer.py:84).  Nearest lidar point of every estimated point (exact
    NN; the reference queries a sklearn KDTree, :180-181), pairs closer than threshold / scale (:189), mean pair distance (:190-194).
    Returns loss [, d loss / d pcd_est, d loss / d scale, nn index, close mask]."""
    est = np.asarray(pcd_est, np.float32)
    lidar = np.asarray(pcd_lidar, np.float32)
    if est.size == 0 or lidar.size == 0:
        return (np.float32(0), np.zeros_like(est), np.float32(0), None, None) if want_grad else np.float32(0)
    fr = (lidar / np.float32(scale)).astype(np.float32)
    d2 = ((est.astype(np.float64)[:, None, :] - fr.astype(np.float64)[None, :, :]) ** 2).sum(-1)
    idx = d2.argmin(1)
    dist = np.sqrt(d2[np.arange(est.shape[0]), idx])
    close = dist < threshold / float(scale)
    if not close.any():
        return (np.float32(0), np.zeros_like(est), np.float32(0), idx, close) if want_grad else np.float32(0)
    diff = fr[idx[close]] - est[close]
    d = np.sqrt((diff * diff).sum(1))
    loss = d.mean(dtype=np.float32)
    if not want_grad:
        return loss
    u = diff / d[:, None] / np.float32(close.sum())               # d loss / d (frustum point) of each pair
    g_est = np.zeros_like(est)
    g_est[close] = -u
    g_scale = np.float32((u * (-lidar[idx[close]] / np.float32(scale) ** 2)).sum())
    return loss, g_est, g_scale, idx, close


def loss_2d(rendering_nocs, css_nocs, diam=5, threshold_nocs=1, want_grad=False):
    """optimizer.py:200-237.  For every rendered pixel (non-zero channel sum, :213) the smallest NOCS distance between its colour and
    the target image weighted by clamp(diam - pixel distance, 0) (:223-226) over ALL pixels (:231-232: outside the window the weighted
    target is 0, so the candidate there is the norm of the rendered colour itself); mean over the pixels whose minimum is below
    threshold_nocs (:233).  Returns loss [, d loss / d rendering_nocs]."""
    r = np.asarray(rendering_nocs, np.float32)
    t = np.asarray(css_nocs, np.float32)
    H, W = r.shape[1:]
    ys, xs = np.nonzero(r.sum(0))
    if (ys.sum() + xs.sum()) == 0:                                   # `if rendering_nonzero_idxs.sum():` -- the SUM OF THE INDICES (:214)
        return (np.float32(0), np.zeros_like(r)) if want_grad else np.float32(0)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    g = np.zeros_like(r)
    dmins, args = [], []
    for y, x in zip(ys, xs):
        w = np.maximum(np.float32(diam) - np.sqrt((yy - np.float32(y)) ** 2 + (xx - np.float32(x)) ** 2), 0).astype(np.float32)
        m = t * w[None]
        v = r[:, y, x]
        diff = np.sqrt(((m - v[:, None, None]) ** 2).sum(0))
        a = int(diff.argmin())
        dmins.append(diff.reshape(-1)[a])
        args.append(a)
    dmins = np.asarray(dmins, np.float32)
    sel = dmins < threshold_nocs
    loss = dmins[sel].mean(dtype=np.float32) if sel.any() else np.float32(np.nan)
    if not want_grad:
        return loss
    cnt = np.float32(sel.sum())
    for (y, x, a, dm, s) in zip(ys, xs, args, dmins, sel):
        if not s:
            continue
        w = max(np.float32(diam) - np.sqrt(np.float32((a // W - y) ** 2 + (a % W - x) ** 2)), np.float32(0))
        m = t[:, a // W, a % W] * w
        g[:, y, x] = (r[:, y, x] - m) / dm / cnt
    return loss, g


# ---- sphere tracing (NOT in the reference: the render mode of BASELINE.json's north_star wording) --------------------
# The oracle of sdflabel_amd.SphereTracer (csrc/trace.hip): the same ray set-up, step rule, hit / exit tests, Newton polish and
# implicit-function gradient, in numpy on a subset of pixels.  There is no reference algorithm to cite; the pose and ray conventions are the
# reference's (pipelines/optimizer.py:86-90 pose, primitives.py:203-208 pixel rays, projection.py:53-55 NOCS colour).

def trace_rays(pose, Kinv, pixels_xy):
    """object-space rays of pixels (n,2): o = -R^T t, d = R^T K^-1 [x, y, 1]; returns o (3,), d (n,3), r_cam (n,3)   (float32)"""
    f = np.float32
    P = np.asarray(pose, f).reshape(4, 4)
    Ki = np.asarray(Kinv, f).reshape(3, 3)
    x, y = np.asarray(pixels_xy, f)[:, 0], np.asarray(pixels_xy, f)[:, 1]
    r = np.stack([Ki[0, 0] * x + Ki[0, 1] * y + Ki[0, 2], Ki[1, 0] * x + Ki[1, 1] * y + Ki[1, 2], Ki[2, 0] * x + Ki[2, 1] * y + Ki[2, 2]], 1).astype(f)
    R, t = P[:3, :3], P[:3, 3]
    d = (r @ R).astype(f)                    # d_j = sum_i R_ij r_i
    o = (-(t @ R)).astype(f)
    return o, d, r


def _trace_slab(o, d, bound, near):
    """slab test of rays o + lam d against the cube [-bound, bound]^3: (l0, l1, enters) -- lam of entry (>= near) and exit"""
    f = np.float32
    n = d.shape[0]
    l0 = np.full(n, near, f)
    l1 = np.full(n, np.finfo(f).max, f)
    active = np.ones(n, bool)
    for a in range(3):
        da = d[:, a]
        par = np.abs(da) < f(1e-12)
        active &= ~par | (np.abs(o[a]) <= bound)
        with np.errstate(divide="ignore", invalid="ignore"):
            ta = ((-f(bound) - o[a]) / da).astype(f)
            tb = ((f(bound) - o[a]) / da).astype(f)
        lo, hi = np.minimum(ta, tb), np.maximum(ta, tb)
        l0 = np.where(par, l0, np.maximum(l0, lo)).astype(f)
        l1 = np.where(par, l1, np.minimum(l1, hi)).astype(f)
    active &= l0 < l1
    return l0, l1, active


def cone_march(layers, spec, latn, pose, Kinv, image_wh, block, blocks, cone_steps=10, eps=2e-3, bound=1.0, near=1e-3, spec_k=1, sigma=0.9):
    """Cone marching of the pixel blocks `blocks` (ids by * nbx + bx of block x block pixel tiles of a W x H image): ONE ray through the centre
    of the (image-clipped) block stands for all its pixels.  All pixel rays share the origin and the parametrisation (lam = camera depth), so
    the block's rays at parameter lam lie within lam * delta of the centre ray's point, delta = max over the block's corner pixels of
    |d_corner - d_centre|.  With v = the distance bound at the centre point: free = v - lam * delta > eps means no surface within the cone's
    cross-section at lam, and the cone may advance by a = free / (|d_c| + delta) (the sphere of radius v around the centre point covers the
    cross-sections up to there).  A cone stops where free <= eps (or NaN): its pixels start their own march at that lam; a cone that advances
    past the far side of the cube for ALL its pixels is culled: none of its pixels can hit; cones still marching after cone_steps passes stop
    where they are.
    The centre point x may lie outside the cube the decoder is defined on (the block's range is the union of its pixels' ranges): the decoder
    is evaluated at c = x clamped into the cube -- no extrapolated value is trusted.  The rendered surface lies inside the cube (rays are clipped
    to it), the cube is convex and c is its nearest point to x, so |x - s|^2 >= cd^2 + |c - s|^2 >= cd^2 + max(f(c), 0)^2 for every surface
    point s: v = sqrt(cd^2 + max(f(c), 0)^2) is a valid distance bound at x (inside the cube cd is exactly 0 and v = f(c)).
    Speculative passes (spec_k > 1; r04): a pass evaluates spec_k samples of a cone at once, p_0 = lam, p_j = p_{j-1} + sigma q^j a_prev
    (a_prev = the cone's previous advance, q = the ratio of its last two advances clamped to [0.5, 1.5]; before the first pass a_prev = 0.1 of the
    block's parameter range).  Sample j counts only inside the range its predecessor proved free, p_{j-1} < p_j <= p_{j-1} + a_{j-1}: the
    accepted prefix is a valid, shorter-stepped cone march; the cone continues from the last accepted sample's full advance.  One cone pass is
    one decoder pass of latency whatever the row count (4096 cones of a 256x256 crop fill a quarter of the chip), so the ~10 sequential
    passes of a cone march become ~4.
    Returns start (per block: lam >= 0 to start from, or -1: culled / no pixel's ray enters the cube), margin (distance of the block's closest
    decision to its threshold), evals."""
    f = np.float32
    W_, H_ = int(image_wh[0]), int(image_wh[1])
    BL = int(block)
    K_ = int(spec_k)
    nbx = (W_ + BL - 1) // BL
    blocks = np.asarray(blocks, np.int64)
    nb = blocks.shape[0]
    x0 = ((blocks % nbx) * BL).astype(np.int64)
    y0 = ((blocks // nbx) * BL).astype(np.int64)
    x1 = np.minimum(x0 + BL - 1, W_ - 1)
    y1 = np.minimum(y0 + BL - 1, H_ - 1)
    cx = (f(0.5) * (x0 + x1).astype(f)).astype(f)
    cy = (f(0.5) * (y0 + y1).astype(f)).astype(f)
    o, dc, _ = trace_rays(pose, Kinv, np.stack([cx, cy], 1))
    delta = np.zeros(nb, f)
    for xs_, ys_ in ((x0, y0), (x1, y0), (x0, y1), (x1, y1)):
        _, dk, _ = trace_rays(pose, Kinv, np.stack([xs_, ys_], 1).astype(f))
        e = (dk - dc).astype(f)
        delta = np.maximum(delta, np.sqrt(e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1] + e[:, 2] * e[:, 2]).astype(f))
    near_b = np.full(nb, np.finfo(f).max, f)
    far_b = np.zeros(nb, f)
    any_in = np.zeros(nb, bool)
    for j in range(BL):
        for i in range(BL):
            px_, py_ = x0 + i, y0 + j
            ins = (px_ <= x1) & (py_ <= y1)
            _, dk, _ = trace_rays(pose, Kinv, np.stack([px_, py_], 1).astype(f))
            l0, l1, act = _trace_slab(o, dk, bound, near)
            act &= ins
            near_b = np.where(act, np.minimum(near_b, l0), near_b).astype(f)
            far_b = np.where(act, np.maximum(far_b, l1), far_b).astype(f)
            any_in |= act
    dn = np.sqrt(dc[:, 0] * dc[:, 0] + dc[:, 1] * dc[:, 1] + dc[:, 2] * dc[:, 2]).astype(f)
    latn = np.asarray(latn, f).reshape(-1)
    L = latn.shape[0]
    start = np.full(nb, -1.0, f)
    margin = np.full(nb, np.inf)
    lam = np.where(any_in, near_b, 0).astype(f)
    aprev = np.where(any_in, (f(0.1) * (far_b - near_b).astype(f)).astype(f), 0).astype(f)
    q = np.ones(nb, f)
    active = any_in.copy()
    evals = 0
    for s in range(int(cone_steps)):
        idx = np.nonzero(active)[0]
        if idx.size == 0:
            break
        m = idx.size
        P = np.zeros((m, K_), f)
        P[:, 0] = lam[idx]
        qp = q[idx].copy()
        for j in range(1, K_):
            P[:, j] = (P[:, j - 1] + ((f(sigma) * qp).astype(f) * aprev[idx]).astype(f)).astype(f)
            qp = (qp * q[idx]).astype(f)
        X = (o[None, None] + P[:, :, None] * dc[idx][:, None, :]).astype(f)
        Xc = np.minimum(np.maximum(X, -f(bound)), f(bound)).astype(f)
        e = (X - Xc).astype(f)
        cd = np.sqrt(((e[..., 0] * e[..., 0]).astype(f) + (e[..., 1] * e[..., 1]).astype(f)).astype(f) + (e[..., 2] * e[..., 2]).astype(f)).astype(f)
        rows = np.concatenate([np.broadcast_to(latn, (m * K_, L)), Xc.reshape(-1, 3)], 1).astype(f)
        v = decoder_forward(layers, spec, rows)[:, 0].astype(f).reshape(m, K_)
        vp = np.maximum(v, f(0))
        v = np.where(cd > 0, np.sqrt(((cd * cd).astype(f) + (vp * vp).astype(f)).astype(f)).astype(f), v).astype(f)
        evals += m * K_
        free = (v - (P * delta[idx, None]).astype(f)).astype(f)
        with np.errstate(invalid="ignore"):
            A = (free / (dn[idx] + delta[idx]).astype(f)[:, None]).astype(f)          # the advance each sample would allow
        ADV = (P + A).astype(f)
        done = np.zeros(m, bool)
        lam_new, a_last, a_before = lam[idx].copy(), aprev[idx].copy(), aprev[idx].copy()
        mg = np.full(m, np.inf)
        for j in range(K_):
            if j > 0:
                gap = (P[:, j] - P[:, j - 1]).astype(f)
                valid = (P[:, j] > P[:, j - 1]) & (gap <= A[:, j - 1])
                mg = np.where(done, mg, np.minimum(mg, np.abs(gap - A[:, j - 1])))
                done |= ~valid                                   # the prediction left the proven range: the pass ends at the previous sample
            take = ~done
            go = free[:, j] > f(eps)                             # (NaN: stop)
            mg = np.where(take, np.minimum(mg, np.abs(free[:, j] - f(eps))), mg)
            stop = take & ~go
            start[idx[stop]] = P[stop, j]
            out = take & go & ~(ADV[:, j] < far_b[idx])
            mg = np.where(take & go, np.minimum(mg, np.abs(ADV[:, j] - far_b[idx])), mg)
            start[idx[out]] = -1.0
            adv_ok = take & go & ~out
            a_before = np.where(adv_ok, a_last, a_before)
            a_last = np.where(adv_ok, A[:, j], a_last)
            lam_new = np.where(adv_ok, ADV[:, j], lam_new)
            active[idx[stop | out]] = False
            done |= stop | out
        margin[idx] = np.minimum(margin[idx], mg)
        keep = active[idx]
        with np.errstate(divide="ignore", invalid="ignore"):
            qn = np.where(a_before > 0, np.minimum(np.maximum((a_last / a_before).astype(f), f(0.5)), f(1.5)), f(1.0)).astype(f)
        lam[idx[keep]] = lam_new[keep]
        aprev[idx[keep]] = a_last[keep]
        q[idx[keep]] = qn[keep]
    start[active] = lam[active]
    return start, margin, evals


def sphere_trace(layers, spec, latn, pose, Kinv, pixels_xy, steps=64, eps=2e-3, bound=1.0, near=1e-3, spec_from=None, spec_k=1, sigma=0.9,
                 cone_block=None, cone_steps=10, image_wh=None, cone_spec_k=1, q_max=1.0):
    """March every ray: x = o + lam d, v = decoder(latn, x); |v| < eps -> hit at lam; else lam += v / |d|; lam >= far (exit of the cube
    [-bound, bound]^3) or NaN -> miss; out of steps -> miss (unresolved).
    Speculative passes (spec_k > 1, from pass index spec_from on): a pass evaluates spec_k samples of the ray at once, p_0 = lam and
    p_j = p_{j-1} + sigma q^j rho / |d| with rho = |v| of the ray's previous accepted sample and q = the ratio of its last two radii (clamped
    to [0.5, 1]) -- a guess of where plain tracing would put its next samples.  Sample j counts only if it lies INSIDE the safe sphere of sample
    j-1 ((p_j - p_{j-1}) |d| <= |v_{j-1}|, v_{j-1} > 0): the accepted prefix is a valid (slightly shorter-stepped) sphere-tracing sequence,
    nothing is skipped; the first accepted sample with |v| < eps is the hit; otherwise the ray continues from the last accepted sample.  Rays
    creeping along a face at grazing incidence -- the ones that keep a march alive for dozens of steps -- advance spec_k samples per pass.
    Which passes are speculative depends on the pass INDEX only, so the sample sequence of a ray is a function of the ray alone.
    cone_block (with image_wh = (W, H)): cone marching first (cone_march above) -- the rays of culled pixel blocks are misses without a single
    evaluation of their own, the others start at max(cube entry, the block's cone stop).
    Then one Newton step along non-grazing rays with the decoder value f0 and input gradient at the marched point: lam_s = lam0 - f0 / (gx . d)
    where |gx . d| > 0.1 |gx| |d|.
    Returns a dict of per-ray arrays: hit (bool), lam0, lam_s, ok (Newton step taken), x_s (n,3), depth, color (NOCS, n,3), normals ((R n + 1)/2,
    n,3), n_hat, gx (n,3), gz (n,L), f0, c (= 1 / (gx . d) or 0), margin (distance of the closest hit / coverage / exit / grazing decision to its
    threshold, in the decision's own units: rays with a small margin may legitimately decide differently under float rounding), n_steps (passes),
    evals (total decoder evaluations, speculative samples included)."""
    f = np.float32
    latn = np.asarray(latn, f).reshape(-1)
    L = latn.shape[0]
    o, d, r = trace_rays(pose, Kinv, pixels_xy)
    n = d.shape[0]
    l0, l1, active = _trace_slab(o, d, bound, near)
    cone_evals = 0
    cone_culled = np.zeros(n, bool)
    cone_margin = np.full(n, np.inf)
    if cone_block:
        W_ = int(image_wh[0])
        BL = int(cone_block)
        nbx = (W_ + BL - 1) // BL
        pxy = np.asarray(pixels_xy, np.int64)
        bid = (pxy[:, 1] // BL) * nbx + pxy[:, 0] // BL
        ub, inv = np.unique(bid, return_inverse=True)
        cstart, cmarg, cone_evals = cone_march(layers, spec, latn, pose, Kinv, image_wh, BL, ub, cone_steps, eps, bound, near, cone_spec_k, sigma)
        cs = cstart[inv]
        cone_margin = cmarg[inv]
        cone_culled = active & (cs < 0)
        l0 = np.where(cs >= 0, np.maximum(l0, cs), l0).astype(f)
        active &= (cs >= 0) & (l0 < l1)
    dn = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(f)
    lam = l0.copy()
    rho, q = np.zeros(n, f), np.ones(n, f)                        # radius of the previous accepted sample, ratio of the last two radii
    hit = np.zeros(n, bool)
    lam0 = np.zeros(n, f)
    margin = cone_margin.copy()
    n_steps = np.zeros(n, np.int32)
    evals = cone_evals
    census = []                                                   # active rays at the start of every pass (schedule studies)
    for s in range(steps):
        idx = np.nonzero(active)[0]
        if idx.size == 0:
            break
        census.append(int(idx.size))
        if isinstance(spec_from, (list, tuple)):                  # a schedule [(first pass index, samples per pass), ...], ascending
            k = 1
            for s_from, s_k in spec_from:
                if s >= s_from:
                    k = int(s_k)
        else:
            k = int(spec_k) if (spec_from is not None and s >= spec_from) else 1
        m = idx.size
        P = np.zeros((m, k), f)
        P[:, 0] = lam[idx]
        qp = q[idx].copy()
        for j in range(1, k):
            P[:, j] = (P[:, j - 1] + ((f(sigma) * qp * rho[idx]).astype(f) / dn[idx]).astype(f)).astype(f)
            qp = (qp * q[idx]).astype(f)
        X = (o[None, None] + P[:, :, None] * d[idx][:, None, :]).astype(f)
        rows = np.concatenate([np.broadcast_to(latn, (m * k, L)), X.reshape(-1, 3)], 1).astype(f)
        V = decoder_forward(layers, spec, rows)[:, 0].astype(f).reshape(m, k)
        evals += m * k
        n_steps[idx] += 1
        R = np.abs(V)
        J = np.zeros(m, np.int64)
        done = R[:, 0] < f(eps)
        ishit = done.copy()
        mg = np.abs(R[:, 0] - eps)
        for j in range(1, k):
            gap = ((P[:, j] - P[:, j - 1]) * dn[idx]).astype(f)
            cov = (gap <= R[:, j - 1]) & (V[:, j - 1] > 0) & (P[:, j] > P[:, j - 1])
            mg = np.where(done, mg, np.minimum(mg, np.abs(gap - R[:, j - 1])))
            take = cov & ~done
            done |= ~cov
            J = np.where(take, j, J)
            hj = take & (R[:, j] < f(eps))
            mg = np.where(take, np.minimum(mg, np.abs(R[:, j] - eps)), mg)
            ishit |= hj
            done |= hj
        margin[idx] = np.minimum(margin[idx], mg)
        ar = np.arange(m)
        pj, vj, rj = P[ar, J], V[ar, J], R[ar, J]
        hit[idx[ishit]] = True
        lam0[idx[ishit]] = pj[ishit]
        prev = np.where(J > 0, R[ar, np.maximum(J - 1, 0)], rho[idx]).astype(f)
        with np.errstate(divide="ignore", invalid="ignore"):
            qn = np.where(prev > 0, np.minimum(np.maximum((rj / prev).astype(f), f(0.5)), f(q_max)), f(1.0)).astype(f)
        l2 = (pj + (vj / dn[idx]).astype(f)).astype(f)
        keep = ~ishit & (l2 < l1[idx]) & ~np.isnan(vj)
        margin[idx[~ishit]] = np.minimum(margin[idx[~ishit]], np.abs(l1[idx[~ishit]] - l2[~ishit]))
        lam[idx[keep]] = l2[keep]
        rho[idx] = rj
        q[idx] = qn
        active[idx] = keep
    unresolved = active.copy()
    out = {"hit": hit, "lam0": lam0, "unresolved": unresolved, "n_steps": n_steps, "evals": evals, "far": l1, "entered": l0 < l1, "cone_culled": cone_culled,
           "cone_evals": cone_evals, "active_per_pass": census}
    hi_ = np.nonzero(hit)[0]
    x0 = (o[None] + lam0[hi_, None] * d[hi_]).astype(f)
    rows = np.concatenate([np.broadcast_to(latn, (hi_.size, L)), x0], 1).astype(f)
    if hi_.size:
        f0h, cache = decoder_forward(layers, spec, rows, want_cache=True)
        Jh = decoder_backward_inputs(layers, spec, rows, cache, np.ones_like(f0h)).astype(f)
        f0h = f0h[:, 0].astype(f)
    else:
        f0h, Jh = np.zeros(0, f), np.zeros((0, L + 3), f)
    gz, gx = np.zeros((n, L), f), np.zeros((n, 3), f)
    f0 = np.zeros(n, f)
    gz[hi_], gx[hi_], f0[hi_] = Jh[:, :L], Jh[:, L:], f0h
    gd = (gx * d).sum(1).astype(f)
    gn = np.sqrt((gx * gx).sum(1)).astype(f)
    ok = hit & (np.abs(gd) > f(0.1) * gn * dn)
    margin[hit] = np.minimum(margin[hit], np.abs(np.abs(gd[hit]) - f(0.1) * gn[hit] * dn[hit]))
    with np.errstate(divide="ignore", invalid="ignore"):
        lam_s = np.where(ok, lam0 - f0 / np.where(ok, gd, 1), lam0).astype(f)
        c = np.where(ok, 1 / np.where(ok, gd, 1), 0).astype(f)
    x_s = (o[None] + lam_s[:, None] * d).astype(f)
    n_hat = (gx / np.maximum(gn, f(1e-12))[:, None]).astype(f)
    R = np.asarray(pose, f).reshape(4, 4)[:3, :3]
    out.update(lam_s=lam_s, ok=ok, c=c, x_s=x_s, gx=gx, gz=gz, f0=f0, n_hat=n_hat, margin=margin,
               depth=np.where(hit, lam_s * r[:, 2], 0).astype(f),
               color=np.where(hit[:, None], (x_s * np.array([-1, 1, 1], f) + 1) / 2, 0).astype(f),
               normals=np.where(hit[:, None], (n_hat @ R.T + 1) / 2, 0).astype(f), r_cam=r, d=d, o=o)
    return out


def sphere_trace_backward(tr, pose, g_color=None, g_depth=None, g_normals=None):
    """Gradient of a functional of the traced images w.r.t. the pose matrix entries and the NORMALISED latent at the fixed hit set, through
        lam(θ) = lam_s - c [ gx . (o(θ) + lam_s d(θ) - x_s) + gz . (z(θ) - z_s) ],   x(θ) = o(θ) + lam(θ) d(θ),   n_cam = R n_hat (n_hat constant)
    tr: the dict sphere_trace returned; g_color / g_normals (n,3), g_depth (n,): upstream gradients of the rays' image values.
    Returns g_pose (4,4) (rotation and translation entries filled) and g_latn (L,)   (float64 sums)."""
    hit = tr["hit"]
    n = hit.shape[0]
    P = np.asarray(pose, np.float64).reshape(4, 4)
    R, t = P[:3, :3], P[:3, 3]
    d, r = tr["d"].astype(np.float64), tr["r_cam"].astype(np.float64)
    gx, gz = tr["gx"].astype(np.float64), tr["gz"].astype(np.float64)
    c, lam_s = tr["c"].astype(np.float64), tr["lam_s"].astype(np.float64)
    z3 = np.zeros((n, 3))
    gxs = z3 if g_color is None else np.asarray(g_color, np.float64) * np.array([-0.5, 0.5, 0.5])
    gnc = z3 if g_normals is None else np.asarray(g_normals, np.float64) * 0.5
    gl = (gxs * d).sum(1) + (0 if g_depth is None else np.asarray(g_depth, np.float64) * r[:, 2])
    k = c * gl
    gw = (gxs - k[:, None] * gx) * hit[:, None]
    go, gdd = gw, lam_s[:, None] * gw
    gnc = gnc * hit[:, None]
    gR = -np.outer(t, go.sum(0)) + r.T @ gdd + gnc.T @ tr["n_hat"].astype(np.float64)
    gt = -(R @ go.sum(0))
    g_pose = np.zeros((4, 4))
    g_pose[:3, :3], g_pose[:3, 3] = gR, gt
    g_latn = -((k * hit)[:, None] * gz).sum(0)
    return g_pose, g_latn


# ---- refinement driven by the sphere tracer (the tracer as a backend of the loop of pipelines/optimizer.py:79-157) -------------------------
# The caller contract is the reference loop's: the renderer hands the loop rendering['color'] (NOCS image -> compute_loss_2d, optimizer.py:132-141)
# and points['xyzf'] (camera-frame surface points -> compute_loss_3d, :125-130); with the tracer those are the traced NOCS image and the hit
# points p_cam = lam_s K^-1 [x, y, 1] of the hit pixels in pixel order.  PARITY UNPINNED (the reference has no tracer); the losses, the pose
# construction, the latent normalisation and the solver are the reference's and pinned by goldens G8 / G12.

def traced_points(tr, H, W):
    """points['xyzf'] of a traced render: camera-frame hit points lam_s * r_cam of the hit pixels, in pixel (row-major) order; tr = the dict of
    sphere_trace over ALL H*W pixels in row-major order.  Returns (xyzf (n_hit,3), pixel index of each row)."""
    hit = tr["hit"]
    assert hit.shape[0] == H * W
    pix = np.nonzero(hit)[0]
    return (tr["lam_s"][pix, None] * tr["r_cam"][pix]).astype(np.float32), pix


def traced_refine_gradients(layers, spec, yaw, trans, scale, latent, K, H, W, target, lidar, w2=0.3, w3=0.5, trace_kwargs=None, points_grad="material", color_grad="material"):
    """One iteration of the loop up to the solver step, rendered by the tracer: returns (weighted loss_2d, weighted loss_3d, g_yaw, g_trans (3,),
    g_scale, g_latent (L,), n_hit) -- the gradients of w3 loss_3d + w2 loss_2d (optimizer.py:144-146) -- or None where the loop skips the frame
    (no hit or no lidar point, :127-129).  Pose optimizer.py:86-90, latent normalisation :96, lidar / scale :84."""
    f = np.float32
    lat = np.asarray(latent, f).reshape(-1)
    nl = max(float(np.sqrt((lat.astype(np.float64) ** 2).sum())), 1e-12)
    latn = (lat / f(nl)).astype(f)
    pose = render_pose(float(np.asarray(yaw).reshape(-1)[0]), np.asarray(trans, f).reshape(3))
    Kinv = np.linalg.inv(np.asarray(K, f)).astype(f)
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    px = np.stack([xs.reshape(-1), ys.reshape(-1)], 1)
    tr = sphere_trace(layers, spec, latn, pose, Kinv, px, image_wh=(W, H), **(trace_kwargs or {}))
    est, pix = traced_points(tr, H, W)
    lidar = np.asarray(lidar, f).reshape(-1, 3)
    if est.shape[0] == 0 or lidar.shape[0] == 0:
        return None
    color = np.ascontiguousarray(tr["color"].T.reshape(3, H, W))
    l2, g_color = loss_2d(color, target, want_grad=True)
    s = float(np.asarray(scale).reshape(-1)[0])
    l3, g_est, g_scale, _, _ = loss_3d(est, lidar, s, want_grad=True)
    gC = (w2 * g_color.astype(np.float64)).reshape(3, -1).T
    ge = w3 * g_est.astype(np.float64)
    if points_grad == "ray":
        # p_cam = lam r with the pixel ray r fixed: only lam moves; sphere_trace_backward takes it as a depth gradient (depth = lam r_z)
        g_depth = np.zeros(H * W, np.float64)
        r = tr["r_cam"].astype(np.float64)
        g_depth[pix] = (ge * r[pix]).sum(1) / r[pix, 2]
        g_pose, g_latn = sphere_trace_backward(tr, pose, g_color=gC, g_depth=g_depth)
    else:
        # material points, the autograd semantics of the reference's points['xyzf'] (grid.py:61 p = x - sdf n_hat with n_hat constant, then
        # projection.py:58 p_cam = R p + t): the hit point x_s moves rigidly with the pose and along its normal with the latent,
        # d x_s = -n_hat (gz . dz) / |gx|
        P64 = pose.astype(np.float64)
        if color_grad == "image":
            g_pose, g_latn = sphere_trace_backward(tr, pose, g_color=gC)
        else:
            # ... and so does the pixel's NOCS colour (projection.py:53-55 colours = the surfel's own object coordinates: pose-independent)
            g_pose = np.zeros((4, 4))
            hitm = tr["hit"]
            gxs = gC * np.array([-0.5, 0.5, 0.5]) * hitm[:, None]
            gna = np.sqrt((tr["gx"].astype(np.float64) ** 2).sum(1))
            kc = -(gxs * tr["n_hat"].astype(np.float64)).sum(1) / np.maximum(gna, 1e-12) * hitm
            g_latn = (kc[:, None] * tr["gz"].astype(np.float64)).sum(0)
        xs = tr["x_s"][pix].astype(np.float64)
        g_pose[:3, :3] += ge.T @ xs
        g_pose[:3, 3] += ge.sum(0)
        go = ge @ P64[:3, :3]                                   # R^T g per point
        gn = np.sqrt((tr["gx"][pix].astype(np.float64) ** 2).sum(1))
        k = -(go * tr["n_hat"][pix].astype(np.float64)).sum(1) / np.maximum(gn, 1e-12)
        g_latn = g_latn + (k[:, None] * tr["gz"][pix].astype(np.float64)).sum(0)
    y = float(np.asarray(yaw).reshape(-1)[0])
    c, sn = np.cos(y), np.sin(y)
    dR = np.array([[-sn, 0, c], [0, 0, 0], [-c, 0, -sn]])                 # d [R_y(yaw) with row 1 negated] / d yaw
    g_yaw = float((g_pose[:3, :3] * dR).sum())
    g_lat = (g_latn - latn.astype(np.float64) * (latn.astype(np.float64) @ g_latn)) / nl
    return f(w2) * l2, f(w3) * l3, g_yaw, g_pose[:3, 3].copy(), float(w3 * g_scale), g_lat, int(est.shape[0])


class TracedRefiner:
    """The reference loop's solver (MultipleOptimizer: Adam lr .01 on yaw and trans, SGD lr .01 on scale and 3e-5 on the latent,
    optimizer.py:13-23,34-52) and skip rules (:127-129,149-151) around traced_refine_gradients; float64 state, one crop."""

    def __init__(self, layers, spec, params, K, H, W, target, lidar, w2=0.3, w3=0.5, trace_kwargs=None, points_grad="material", color_grad="material"):
        self.points_grad, self.color_grad = points_grad, color_grad
        self.layers, self.spec, self.K, self.H, self.W = layers, spec, np.asarray(K, np.float32), int(H), int(W)
        self.target, self.lidar, self.w2, self.w3 = np.asarray(target, np.float32), np.asarray(lidar, np.float32), w2, w3
        self.kw = trace_kwargs or {}
        self.p = np.concatenate([np.asarray(params[k], np.float64).reshape(-1) for k in ("yaw", "trans", "scale", "latent")])
        self.m, self.v, self.t = np.zeros(4), np.zeros(4), 0
        self.log = []

    def step(self):
        p = self.p.astype(np.float32)
        out = traced_refine_gradients(self.layers, self.spec, p[0:1], p[1:4], p[4:5], p[5:], self.K, self.H, self.W, self.target, self.lidar,
                                      self.w2, self.w3, self.kw, self.points_grad, self.color_grad)
        if out is None:
            return False
        l2, l3, g_yaw, g_trans, g_scale, g_lat, n_hit = out
        total = float(l2) + float(l3)
        if np.isnan(total) or total == 0:
            return False
        self.log.append((float(l2), float(l3), n_hit))
        g = np.concatenate([[g_yaw], g_trans])
        self.t += 1
        b1, b2 = 0.9, 0.999
        self.m = b1 * self.m + (1 - b1) * g
        self.v = b2 * self.v + (1 - b2) * g * g
        denom = np.sqrt(self.v) / np.sqrt(1 - b2 ** self.t) + 1e-8
        self.p[0:4] -= (0.01 / (1 - b1 ** self.t)) * (self.m / denom)
        self.p[4] -= 0.01 * g_scale
        self.p[5:] -= 0.00003 * g_lat
        return True
