"""Shared test helpers."""
import torch


def bare_model(D=128):
    from hvlm.config import HandsOnVLMConfig
    from hvlm.model import HandsOnVLMForCausalLM
    m = object.__new__(HandsOnVLMForCausalLM)
    m.config = HandsOnVLMConfig(hidden_size=D, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2)
    m.device = torch.device("cpu")
    m.token_dim = D
    return m


