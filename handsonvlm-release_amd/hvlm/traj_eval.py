"""Trajectory metrics ADE / FDE / WDE (numpy), mirror of hoi_forecast/evaluation/traj_eval.py.

`evaluate_traj_stochastic` keeps the reference's return convention by default: it returns
(FDE, ADE, WDE) under the names (ade, fde, wde) because the reference feeds compute_fde into the
`ade` list and compute_ade into the `fde` list (traj_eval.py:187,190).  Pass
`reference_labels=False` for correctly named values.  Unlike the reference's eval loop
(handsonvlm_inference.py:173) the metric is meant to be called once over all samples, not O(N^2).
"""
from __future__ import annotations

import numpy as np


def _in_frame(gt):
    return (gt[..., 0] >= 0) & (gt[..., 1] >= 0) & (gt[..., 0] < 1) & (gt[..., 1] < 1)


def compute_ade(pred_traj, gt_traj, valid_traj=None, reduction=True):
    """pred/gt [B,N,T,2], valid [B,N]: mean over time of the point error; out-of-frame GT points count 0."""
    pred_traj, gt_traj = np.asarray(pred_traj), np.asarray(gt_traj)
    if valid_traj is None:
        valid_traj = np.ones(pred_traj.shape[:2], dtype=int)
    err = (gt_traj - pred_traj) * _in_frame(gt_traj)[..., None]
    ade = np.linalg.norm(err, axis=3).mean(axis=2) * valid_traj
    if reduction:
        return ade.sum() / valid_traj.sum(), valid_traj.sum()
    return ade, valid_traj


def compute_fde(pred_traj, gt_traj, valid_traj=None, reduction=True):
    pred_traj, gt_traj = np.asarray(pred_traj), np.asarray(gt_traj)
    if valid_traj is None:
        valid_traj = np.ones(pred_traj.shape[:2], dtype=int)
    gl, pl = gt_traj[:, :, -1, :], pred_traj[:, :, -1, :]
    err = (gl - pl) * _in_frame(gl)[..., None]
    fde = np.linalg.norm(err, axis=2) * valid_traj
    if reduction:
        return fde.sum() / valid_traj.sum(), valid_traj.sum()
    return fde, valid_traj


def compute_wde(pred_traj, gt_traj, valid_traj=None, reduction=True):
    pred_traj, gt_traj = np.asarray(pred_traj), np.asarray(gt_traj)
    if valid_traj is None:
        valid_traj = np.ones(pred_traj.shape[:2])
    ok = _in_frame(gt_traj)
    w = np.arange(1, 5) / 4
    dist = np.sqrt((((gt_traj - pred_traj) * ok[..., None]) ** 2).sum(-1))
    wde = ((dist * w) * ok).sum(-1) / np.maximum(ok.sum(-1), 1) * valid_traj
    if reduction:
        return wde.sum() / (valid_traj.sum() + 1e-6), valid_traj.sum()
    return wde, valid_traj


def evaluate_traj_stochastic(preds, gts, valids, reference_labels: bool = True):
    """preds [N,K,2,4,2], gts [N,2,4,2], valids [N,2] -> (ade_mean, fde_mean, wde_mean)."""
    preds, gts, valids = np.asarray(preds), np.asarray(gts), np.asarray(valids)
    N, K, n_obj = preds.shape[:3]
    ade = np.array([compute_ade(preds[:, i], gts, valids, False)[0] for i in range(K)], dtype=np.float32)
    fde = np.array([compute_fde(preds[:, i], gts, valids, False)[0] for i in range(K)], dtype=np.float32)
    wde = np.array([compute_wde(preds[:, i], gts, valids, False)[0] for i in range(K)], dtype=np.float32)
    sel = valids > 0
    ade_mean = ade[:, sel].mean()
    fde_mean = fde[:, sel].mean()
    wde_mean = wde.mean()
    # the reference's own consistency check (traj_eval.py:250-253)
    assert np.isclose((ade.mean(0) * valids).sum() / valids.sum(), ade_mean, atol=1e-3)
    if reference_labels:
        return fde_mean, ade_mean, wde_mean
    return ade_mean, fde_mean, wde_mean


def evaluate_traj(preds, gts, valids, reference_labels: bool = True):
    """Deterministic variant (traj_eval.py:74-93), same label swap."""
    ade, _ = compute_ade(preds, gts, valids)
    fde, _ = compute_fde(preds, gts, valids)
    return (fde, ade) if reference_labels else (ade, fde)
