"""Optimizer -- same constructor and `optimize` call as the reference's refinement loop object (pipelines/optimizer.py:43-164), run by
the device-resident BatchRefiner (one crop): the caller (pipelines/refine_css_demo.py:157-191, refine_css.py) keeps building its `params`
dict, constructing `Optimizer(params, device, weights)` and calling `optimize(...)`, then reads the refined `params[...]` tensors.

What differs from the reference is only how an iteration executes: no per-iteration host work (the lidar cloud is uploaded once, the 3-D
nearest-neighbour loss and the 2-D NOCS window loss run in HIP kernels, the Adam/SGD update and the reference's skip rules
(optimizer.py:127-129,149-151) are one kernel), and the iterations are replayed from a HIP graph.  The arithmetic per iteration is the
reference's (trajectory-tested against its own Optimizer, golden G8).  `get_opt_params` (optimizer.py:26-40) turns the caller's arrays into
float32 leaf tensors in place; so does this class, and after `optimize` those tensors hold the refined values.
"""
import weakref

import numpy as np
import torch

from .. import _lib
from ..grid import Grid3D
from ..refine import BatchRefiner


# Refiners (device buffers + captured HIP graph) are shared by all Optimizer objects of a process: the reference's callers construct a new
# Optimizer per crop (refine_css.py:203), and re-allocating / re-capturing for each would cost more than refining it.  Keyed by everything
# the buffers and the captured launches depend on; a few entries at most.
_REFINERS = {}
_REFINERS_MAX = 4
STATS = {"refiners_built": 0}     # (bench / tests)


def clear_refiner_cache():
    """Release the cached BatchRefiners (their device buffers, HIP graphs and the references they hold to the decoders).  Optimizer objects
    created afterwards build fresh ones; existing Optimizer objects keep working with the refiner they already hold."""
    _REFINERS.clear()


def get_opt_params(params, device):
    """optimizer.py:26-40: every entry of `params` becomes a float32 leaf tensor on `device` (in place); returns the parameter groups with
    the reference's learning rates (yaw .01, trans .01 -> Adam; scale .01, latent 3e-5 -> SGD)."""
    for key, value in params.items():
        params[key] = torch.as_tensor(np.asarray(value.detach().cpu() if torch.is_tensor(value) else value, dtype=np.float32)).to(device).requires_grad_(True)
    groups = [{'params': params['yaw'], 'lr': 0.01}, {'params': params['trans'], 'lr': 0.01},
              {'params': params['scale'], 'lr': 0.01}, {'params': params['latent'], 'lr': 0.00003}]
    return params, groups


def _as_np(a):
    return np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float32)


def _grid_density(grid):
    G = int(grid.points.size(0))
    D = int(round(G ** (1.0 / 3.0)))
    if D ** 3 != G:
        raise _lib.SdfrError("grid of %d points is not a D^3 Grid3D" % G)
    return D


def _check_grid(rf, grid, D):
    # the caller's grid must be the Grid3D(D) point set the kernels index, in whatever precision it was built (the reference's callers pass
    # Grid3D(grid_density, device, precision) with the config's float16 default, refine_css.py:148): compare in the caller's dtype -- the
    # same float32 -> precision rounding produced both.  Once per (refiner, grid object): a frame's annotations share one grid.
    seen = getattr(rf, "_grids_ok", None)
    if seen is None:
        seen = rf._grids_ok = weakref.WeakSet()
    if grid in seen:
        return
    pts = grid.points.detach()
    if rf.br is not None and (pts.shape != rf.br.grid.shape or not torch.equal(rf.br.grid.to(pts.dtype), pts.to(rf.br.grid.device))):
        raise _lib.SdfrError("grid.points is not the Grid3D(%d) point set the kernels index" % D)
    seen.add(grid)


def optimize_many(annotations, iters_optim, dsdf, grid, device, weights, render='splat', trace_grad='surfel', tracer_kwargs=None,
                  candidate_reuse=True, max_batch=64, optimize_latent=True):
    """Frame-level entry (r06): refine ALL annotations of a frame together instead of one `Optimizer(...).optimize(...)` call per annotation
    (pipelines/refine_css.py:94,203-223 loops over a frame's annotations one at a time).  Same arithmetic per annotation -- a crop refined here
    is bit-identical to the same crop refined alone through `Optimizer.optimize` (GPU test) -- but the annotations share every launch of an
    iteration (ragged extents: each keeps its own crop size and intrinsics), so a frame of 8-16 cars costs little more than one.

    annotations: list of (params, nocs_pred, pcd_frustum_np, K, crop_size) tuples, or dicts with those keys -- per annotation exactly what the
                 reference passes to Optimizer(params, ...) and optimize(iters, nocs_pred, pcd_frustum_np, dsdf, grid, K, crop_size).
    Each `params` dict is turned into float32 leaf tensors on `device` in place (get_opt_params, optimizer.py:26-40) and holds the refined
    values afterwards; the list of those dicts is returned.  Every annotation starts with fresh solver state, as a new Optimizer object would
    (optimizer.py:47-52).  More than `max_batch` annotations are processed in chunks."""
    if render not in ('splat', 'trace'):
        raise ValueError("render must be 'splat' or 'trace'")
    items = []
    for a in annotations:
        if isinstance(a, dict):
            a = (a['params'], a['nocs_pred'], a['pcd_frustum_np'], a['K'], a['crop_size'])
        params, nocs, lidar, K, crop_size = a
        params, _ = get_opt_params(params, device)
        items.append((params, torch.as_tensor(nocs, dtype=torch.float32), np.asarray(lidar, dtype=np.float32).reshape(-1, 3), _as_np(K).reshape(3, 3),
                      (int(crop_size[0]), int(crop_size[1]))))
    if not items:
        return []
    D = _grid_density(grid)
    dev = grid.points.device
    tracer_kwargs = dict(tracer_kwargs or {})
    n_lidar = max(it[2].shape[0] for it in items)
    cap = max(1024, 1 << (max(n_lidar, 1) - 1).bit_length())
    pmax = max(1024, 1 << (max(it[4][0] * it[4][1] for it in items) - 1).bit_length())
    longest = max(max(it[4]) for it in items)
    while 4 * int(np.ceil(np.sqrt(pmax))) < longest:                  # (a crop more elongated than 16:1: a larger pixel capacity carries its side)
        pmax *= 4
    side = 4 * int(np.ceil(np.sqrt(pmax)))
    max_batch = max(1, int(max_batch))
    for c0 in range(0, len(items), max_batch):
        chunk = items[c0:c0 + max_batch]
        n = len(chunk)
        B = min(max_batch, 1 << (n - 1).bit_length())                # refiners are built per power-of-two batch size; a short batch is padded
        key = ('many', id(dsdf), D, pmax, cap, B, str(dev), dsdf._param_key(dev), float(weights.get('2d', 0.3)), float(weights.get('3d', 0.5)),
               getattr(dsdf, 'mlp_precision', None), bool(optimize_latent), render, trace_grad, tuple(sorted(tracer_kwargs.items())),
               bool(candidate_reuse))
        hit = _REFINERS.get(key)
        if hit is not None and hit[0]() is dsdf:
            rf = hit[1]
        else:
            h0, w0 = chunk[0][4]
            rf = BatchRefiner(dsdf, D, chunk[0][3], (h0, w0), B, lidar_cap=cap, weights=weights, device=dev, optimize_latent=optimize_latent,
                              render=render, trace_grad=trace_grad, tracer_kwargs=tracer_kwargs, max_pixels=pmax, max_side=side,
                              candidate_reuse=bool(candidate_reuse))
            STATS["refiners_built"] += 1
            while len(_REFINERS) >= _REFINERS_MAX:
                _REFINERS.pop(next(iter(_REFINERS)))
            _REFINERS[key] = (weakref.ref(dsdf), rf)
        _check_grid(rf, grid, D)
        sel = list(range(n)) + [n - 1] * (B - n)                     # padding: copies of the last annotation (their rows are dropped)
        with torch.no_grad():
            P = {k: torch.stack([chunk[i][0][k].detach().reshape(-1) for i in sel]) for k in ('yaw', 'trans', 'scale', 'latent')}
            rf.set_crops(P, [chunk[i][1] for i in sel], [chunk[i][2] for i in sel], K=np.stack([chunk[i][3] for i in sel]),
                         crop_sizes=[chunk[i][4] for i in sel])
            if iters_optim > 3 and rf._replay is None:
                rf.capture()
            rf.optimize(iters_optim)
            rf.check_overflow()
            for i in range(n):
                p = chunk[i][0]
                p['yaw'].copy_(rf.yaw[i].view_as(p['yaw']))
                p['trans'].copy_(rf.trans[i].view_as(p['trans']))
                p['scale'].copy_(rf.scale[i].view_as(p['scale']))
                p['latent'].copy_(rf.latent[i].view_as(p['latent']))
    return [it[0] for it in items]


class Optimizer:
    optimize_many = staticmethod(optimize_many)        # Optimizer.optimize_many(annotations, iters, dsdf, grid, device, weights)

    def __init__(self, params, device, weights, rot='dcm', render='splat', trace_grad='surfel', tracer_kwargs=None, candidate_reuse=True):
        """render='trace' (extension): the loop's renderer is the sphere tracer instead of the reference's surfel splat -- same losses, same
        solver, same call (BatchRefiner(render='trace')); not the reference's algorithm, so no parity claim goes with it.
        candidate_reuse (float16 decoders, the reference's shipped precision; r05): the half decoder runs on the band candidates alone while a
        proven Lipschitz bound keeps the candidate set valid -- the same bits as evaluating the whole grid every iteration (GPU tests), audited
        at run time; False evaluates all G rows every iteration as the reference does."""
        self.candidate_reuse = bool(candidate_reuse)
        if render not in ('splat', 'trace'):
            raise ValueError("render must be 'splat' or 'trace'")
        self.render, self.trace_grad, self.tracer_kwargs = render, trace_grad, dict(tracer_kwargs or {})
        if rot != 'dcm':
            raise NotImplementedError("the refinement loop optimises a yaw angle (rot='dcm', optimizer.py:44,86-90); the quaternion "
                                      "variant is commented out in the reference (optimizer.py:92-93)")
        self.params, self.optim_params = get_opt_params(params, device)
        self.weights = weights
        self.rot = rot
        self.log = []             # per-iteration (weighted 2-D loss, weighted 3-D loss, total) when optimize(..., verbose=True)
        self._refiner = None
        self._key = None
        self._adam = None         # Adam moments / step count of yaw and trans: the reference creates its solver in __init__ (optimizer.py:47-52),
                                  # so the state carries over from one optimize() call of an Optimizer object to the next

    def _refiner_for(self, dsdf, grid, K, crop_size, n_lidar, optimize_latent=True):
        D = _grid_density(grid)
        dev = grid.points.device
        cap = max(1024, 1 << (max(n_lidar, 1) - 1).bit_length())      # lidar capacity, rounded up so that a refiner is reused across crops
        Kn = _as_np(K)
        # ragged extents -- the refiner is keyed on a pixel CAPACITY (next power of two of the crop's area), not on the crop's
        # size or intrinsics, which reach the kernels as data (set_crops): the pipeline's crops all have their own (H, W) and K
        # (utils/refinement.py:586-609), yet share one set of buffers and one captured graph -- with either renderer.
        ragged = True
        H_, W_ = int(crop_size[0]), int(crop_size[1])
        pmax = max(1024, 1 << (max(H_ * W_, 1) - 1).bit_length())
        side = 4 * int(np.ceil(np.sqrt(pmax)))
        if ragged and max(H_, W_) > side:
            ragged = False                                            # (a crop more elongated than 16:1: its own fixed-size refiner)
        shape_key = (pmax,) if ragged else (H_, W_, Kn.tobytes())
        key = (id(dsdf), D, shape_key, cap, str(dev), dsdf._param_key(dev),
               float(self.weights.get('2d', 0.3)), float(self.weights.get('3d', 0.5)), getattr(dsdf, 'mlp_precision', None), bool(optimize_latent),
               self.render, self.trace_grad, tuple(sorted(self.tracer_kwargs.items())), self.candidate_reuse)
        if self._key != key:
            hit = _REFINERS.get(key)
            if hit is not None and hit[0]() is dsdf:
                rf = hit[1]
            else:
                rf = BatchRefiner(dsdf, D, Kn, crop_size, 1, lidar_cap=cap, weights=self.weights, device=dev, optimize_latent=optimize_latent,
                                  render=self.render, trace_grad=self.trace_grad, tracer_kwargs=self.tracer_kwargs,
                                  max_pixels=pmax if ragged else None, max_side=side if ragged else None, candidate_reuse=self.candidate_reuse)
                STATS["refiners_built"] += 1
                while len(_REFINERS) >= _REFINERS_MAX:
                    _REFINERS.pop(next(iter(_REFINERS)))
                _REFINERS[key] = (weakref.ref(dsdf), rf)
            _check_grid(rf, grid, D)
            self._refiner, self._key = rf, key
        return self._refiner

    def optimize(self, iters_optim, nocs_pred, pcd_frustum_np, dsdf, grid, K, crop_size, viz_type=None, frame_vis=None, verbose=False,
                 optimize_latent=True):
        """optimizer.py:56-164.  nocs_pred (3,h,w) CSS prediction, pcd_frustum_np (M,3) lidar points of the frustum (camera frame),
        dsdf the decoder, grid a Grid3D, K (3,3), crop_size (H,W).  viz_type must be None (visualisation is not on the path).
        verbose=True prints the reference's per-iteration loss line (one host synchronisation per iteration, as the reference has).
        optimize_latent=False (extension): pose-only refinement -- the latent group gets no update and the shape is evaluated once."""
        if viz_type is not None:
            raise NotImplementedError("visualisation (open3d / matplotlib, optimizer.py:75-77,158-163) is outside the renderer path; "
                                      "call with viz_type=None")
        lidar = np.asarray(pcd_frustum_np, dtype=np.float32).reshape(-1, 3)
        rf = self._refiner_for(dsdf, grid, K, crop_size, lidar.shape[0], optimize_latent)
        p = self.params
        with torch.no_grad():
            rf.set_crops({'yaw': p['yaw'].detach().reshape(1, -1), 'trans': p['trans'].detach().reshape(1, 3),
                          'scale': p['scale'].detach().reshape(1, -1), 'latent': p['latent'].detach().reshape(1, -1)},
                         torch.as_tensor(nocs_pred, dtype=torch.float32)[None], [lidar],
                         **({'K': np.asarray(K.detach().cpu() if torch.is_tensor(K) else K, dtype=np.float32),
                             'crop_sizes': [(int(crop_size[0]), int(crop_size[1]))]} if rf.ragged else {}))
            if self._adam is not None:                        # refiners are shared between Optimizer objects; the solver state is not
                rf.adam_m.copy_(self._adam[0]); rf.adam_v.copy_(self._adam[1]); rf.adam_t.copy_(self._adam[2])
            self.log = []
            if verbose:
                for e in range(iters_optim):
                    rf.iteration()
                    l2, l3 = float(rf.loss2d[0]) * rf.w2, float(rf.loss3d[0]) * rf.w3
                    if int(rf.stepped[0]):
                        print('ITER {} | Losses: 2D - {}, 3D - {}, Total - {}'.format(e, l2, l3, l2 + l3))
                    else:
                        print('Skip frame')
                    self.log.append((l2, l3, l2 + l3))
            else:
                if iters_optim > 3 and rf._replay is None:
                    rf.capture()                              # once per refiner: later crops replay the same graph
                rf.optimize(iters_optim)
            rf.check_overflow()                               # a truncated band must not pass silently (one sync, after the loop)
            self._adam = (rf.adam_m.clone(), rf.adam_v.clone(), rf.adam_t.clone())
            p['yaw'].copy_(rf.yaw.view_as(p['yaw']))
            p['trans'].copy_(rf.trans.view_as(p['trans']))
            p['scale'].copy_(rf.scale.view_as(p['scale']))
            p['latent'].copy_(rf.latent.view_as(p['latent']))
        return self.params
