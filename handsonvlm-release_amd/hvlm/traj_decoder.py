"""Hand-trajectory head (stays PyTorch-ROCm host code, per BASELINE.json north_star).

Mirror of handsonvlm/model/language_model/traj_decoder.py (TrajDecoder / CVAETrajDecoder /
MLPTrajDecoder) over hoi_forecast/architecture/traj_decoder.py (TrajCVAE, TrajMLP) and
hoi_forecast/architecture/decoder_modules.py (VAE).  Module / parameter names match the reference's
state dict (`hand_traj_decoder.hand_traj_decoder.cvae.{enc_MLP.0,linear_means,linear_log_var,
dec_MLP.0,dec_MLP.2}`), so reference checkpoints load unchanged.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class VAE(nn.Module):
    """Conditional VAE over 2-D points: encoder MLP -> (mean, log_var) -> z -> decoder MLP
    (decoder_modules.py:6-61)."""

    def __init__(self, in_dim: int, hidden_dim: int, latent_dim: int, condition_dim: int):
        super().__init__()
        self.in_dim, self.latent_dim, self.condition_dim = in_dim, latent_dim, condition_dim
        self.enc_MLP = nn.Sequential(nn.Linear(in_dim + condition_dim, hidden_dim), nn.ELU())
        self.linear_means = nn.Linear(hidden_dim, latent_dim)
        self.linear_log_var = nn.Linear(hidden_dim, latent_dim)
        self.dec_MLP = nn.Sequential(nn.Linear(latent_dim + condition_dim, hidden_dim), nn.ELU(),
                                     nn.Linear(hidden_dim, in_dim))

    def forward(self, gt: torch.Tensor, condition: torch.Tensor, eps: torch.Tensor | None = None):
        h = self.enc_MLP(torch.cat((gt, condition), dim=-1))
        mean, log_var = self.linear_means(h), self.linear_log_var(h)
        std = torch.exp(0.5 * log_var)
        if eps is None:
            eps = torch.randn_like(std)            # reparameterize (decoder_modules.py:50-53)
        z = (mean + eps * std).to(gt.dtype)
        rec = self.dec_MLP(torch.cat((z, condition), dim=-1))
        recon_loss = torch.sum((rec - gt) ** 2, dim=1)
        kld = -0.5 * torch.sum(1 + log_var - mean.pow(2) - log_var.exp(), dim=1)
        return rec, recon_loss, kld

    def inference(self, z: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
        return self.dec_MLP(torch.cat((z, c), dim=-1))


class TrajCVAE(nn.Module):
    def __init__(self, token_dim: int, hidden_dim: int = 512, latent_dim: int = 256, z_scale: float = 2.0,
                 cond_dtype: torch.dtype | None = torch.bfloat16):
        super().__init__()
        self.token_dim, self.latent_dim, self.z_scale = token_dim, latent_dim, z_scale
        self.cond_dtype = cond_dtype
        self.cvae = VAE(in_dim=2, hidden_dim=hidden_dim, latent_dim=latent_dim, condition_dim=token_dim)

    def forward(self, hand_embedding, shifted_gt_hand, future_valid, eps=None):
        """hand_embedding [B*2*T, C], shifted_gt_hand [B*2*T, 2], future_valid [B,2]
        -> (pred [B*2*T,2], traj_loss [B], KLD [B])  (hoi_forecast/architecture/traj_decoder.py:31-70)."""
        B = future_valid.shape[0]
        T = hand_embedding.shape[0] // (B * 2)
        cond = hand_embedding
        if self.cond_dtype is not None:
            cond = cond.to(self.cond_dtype)        # the reference force-casts to bf16 (:56)
        pred, recon, kld = self.cvae(shifted_gt_hand, cond, eps)
        fv = future_valid.to(recon.dtype)
        traj = (recon.reshape(B, 2, T).sum(-1) * fv).sum(1)
        # parity quirk (:61-66): KLD is not reshaped before .sum(-1) -> scalar sum over the whole batch,
        # then broadcast against future_valid.
        kl = (kld.sum(-1) * fv).sum(1)
        return pred, traj, kl

    def inference(self, hand_embedding: torch.Tensor, z: torch.Tensor | None = None) -> torch.Tensor:
        n = hand_embedding.shape[0]
        if z is None:
            z = self.z_scale * torch.randn([n, self.latent_dim], device=hand_embedding.device).to(hand_embedding.dtype)
        return self.cvae.inference(z, c=hand_embedding)


class TrajMLP(nn.Module):
    """Deterministic ablation head (hoi_forecast/architecture/traj_decoder.py:95-151)."""

    def __init__(self, token_dim: int, hidden_dim: int = 512):
        super().__init__()
        self.token_dim = token_dim
        self.mlp = nn.Sequential(nn.Linear(token_dim, hidden_dim), nn.ReLU(inplace=True),
                                 nn.Linear(hidden_dim, hidden_dim), nn.ReLU(inplace=True), nn.Linear(hidden_dim, 2))

    def forward(self, hand_embedding, shifted_gt_hand, gt_hand_valid, eps=None):
        B = gt_hand_valid.shape[0]
        T = hand_embedding.shape[0] // (B * 2)
        pred = self.mlp(hand_embedding)
        recon = ((pred - shifted_gt_hand) ** 2).sum(-1).reshape(B, 2, T)
        recon = torch.sum(recon * gt_hand_valid) / (torch.sum(gt_hand_valid) + 1e-6)
        return pred, recon.repeat(B), torch.zeros(B, device=hand_embedding.device)

    def inference(self, hand_embedding, z=None):
        return self.mlp(hand_embedding)


class TrajDecoder(nn.Module):
    def __init__(self, token_dim: int):
        super().__init__()
        self.token_dim = token_dim
        self.hand_traj_decoder: nn.Module = None

    def forward(self, pred_hand_embeddings, future_hands, future_valid, lambda_traj, lambda_traj_kl, eps=None):
        """pred_hand_embeddings [B,2,4,C], future_hands [B,2,4,2], future_valid [B,2] -> loss dict
        (handsonvlm traj_decoder.py:12-37: sums over the batch, not means)."""
        B = pred_hand_embeddings.shape[0]
        assert pred_hand_embeddings.shape == (B, 2, 4, self.token_dim), pred_hand_embeddings.shape
        assert future_hands.shape == (B, 2, 4, 2) and future_valid.shape == (B, 2)
        _, traj, kl = self.hand_traj_decoder(pred_hand_embeddings.reshape(-1, self.token_dim),
                                             future_hands.reshape(B * 8, 2), future_valid, eps)
        loss = {"traj_loss": lambda_traj * traj.sum(), "traj_kl_loss": lambda_traj_kl * kl.sum()}
        loss["total_loss"] = loss["traj_loss"] + loss["traj_kl_loss"]
        return loss

    def inference(self, pred_hand_embeddings, z=None):
        B, _, T, _ = pred_hand_embeddings.shape
        out = self.hand_traj_decoder.inference(pred_hand_embeddings.reshape(-1, self.token_dim), z)
        return out.reshape(B, 2, T, 2)


class CVAETrajDecoder(TrajDecoder):
    def __init__(self, token_dim: int):
        super().__init__(token_dim)
        self.hand_traj_decoder = TrajCVAE(token_dim=token_dim, hidden_dim=512, latent_dim=256)


class MLPTrajDecoder(TrajDecoder):
    def __init__(self, token_dim: int):
        super().__init__(token_dim)
        self.hand_traj_decoder = TrajMLP(token_dim=token_dim, hidden_dim=512)
