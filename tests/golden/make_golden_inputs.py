"""Seeded inputs shared by the fixture generator (`make_golden.py`) and the tests, so the
fixtures only need to store expected OUTPUTS.  No reference code here."""
import torch

PROMPT_PAIRS = [
    ("a photo of a house on a mountain", "a photo of a house on a mountain at fall"),  # edit_syn.py defaults
    ("a gray horse in the field", "a whie horse in the field"),  # edit_real.py defaults (typo is the reference's)
    ("a cat sitting on a bench", "a dog sitting on a bench"),
    ("a cat sitting on a bench", "a cat sitting on a wooden bench"),
    ("a extraordinarily fluffy cat", "a fluffy cat"),
    ("photo of a cat riding on a bicycle", "photo of a cat riding on a motorcycle"),
    ("a bowl of fruit", "a bowl of strawberries and fruit on the table"),
    ("soup", "pea soup"),
    ("the quick brown fox jumps", "the quick red fox leaps"),
    ("a b c d e f g", "a c e g"),
    ("children drawing of a castle next to a river", "children drawing of a castle next to a river"),
    ("interchangeable characteristics", "interchangeable words"),
]


def softmax_maps(seed, bh, n, l):
    """row-stochastic maps [bh, n, l], deterministic on the CPU generator"""
    g = torch.Generator().manual_seed(seed)
    return torch.softmax(torch.randn(bh, n, l, generator=g) * 2.0, dim=-1)
