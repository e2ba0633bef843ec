"""Host-side token alignment between the source and target prompt (integer work, bit-exact).

Mirrors the public functions of `/root/reference/p2p/model/seq_aligner.py` (same names,
arguments and return types) so controllers written against the reference keep working:

  get_refinement_mapper  (:121-128)  -> (mapper int64 [P-1, 77], alphas fp32 [P-1, 77])
  get_replacement_mapper (:188-195)  -> fp32 [P-1, 77, 77]
  get_word_inds          (:131-149)
  get_equalizer          (:197-207)

The alignment is Needleman-Wunsch with gap 0 / match +1 / mismatch -1 and the reference's
tie-breaking (prefer "left", then "up", then "diagonal", :70-76), written here as one
dynamic-programming pass over Python ints.  Checked against outputs of the reference module
in `tests/golden/p2p_host.npz`.
"""
from typing import List, Sequence, Tuple, Union

import numpy as np
import torch

GAP, MATCH, MISMATCH = 0, 1, -1
_LEFT, _UP, _DIAG = 1, 2, 3


def _align_pairs(x: Sequence[int], y: Sequence[int]) -> List[Tuple[int, int]]:
    """For every position j of y (ascending): (j, i) with i the aligned position in x or -1."""
    nx, ny = len(x), len(y)
    score = [[0] * (ny + 1) for _ in range(nx + 1)]
    move = [[0] * (ny + 1) for _ in range(nx + 1)]
    for j in range(1, ny + 1):
        score[0][j] = j * GAP
        move[0][j] = _LEFT
    for i in range(1, nx + 1):
        score[i][0] = i * GAP
        move[i][0] = _UP
    for i in range(1, nx + 1):
        xi = x[i - 1]
        row, above, mrow = score[i], score[i - 1], move[i]
        for j in range(1, ny + 1):
            left = row[j - 1] + GAP
            up = above[j] + GAP
            diag = above[j - 1] + (MATCH if xi == y[j - 1] else MISMATCH)
            best = max(left, up, diag)
            row[j] = best
            mrow[j] = _LEFT if best == left else (_UP if best == up else _DIAG)
    pairs = []
    i, j = nx, ny
    while i > 0 or j > 0:
        m = move[i][j]
        if m == _DIAG:
            i, j = i - 1, j - 1
            pairs.append((j, i))
        elif m == _LEFT:
            j -= 1
            pairs.append((j, -1))
        else:
            i -= 1
    pairs.reverse()
    return pairs


def get_mapper(x: str, y: str, tokenizer, max_len: int = 77):
    xs, ys = tokenizer.encode(x), tokenizer.encode(y)
    pairs = _align_pairs(xs, ys)
    n = len(pairs)
    src = torch.tensor([p[1] for p in pairs], dtype=torch.int64)
    alphas = torch.ones(max_len)
    alphas[:n] = src.ne(-1).float()
    mapper = torch.zeros(max_len, dtype=torch.int64)
    mapper[:n] = src
    mapper[n:] = len(ys) + torch.arange(max_len - len(ys))
    return mapper, alphas


def get_refinement_mapper(prompts, tokenizer, max_len: int = 77):
    out = [get_mapper(prompts[0], p, tokenizer, max_len) for p in prompts[1:]]
    return torch.stack([m for m, _ in out]), torch.stack([a for _, a in out])


def get_word_inds(text: str, word_place: Union[int, str], tokenizer) -> np.ndarray:
    """Token positions (1-based: BOS is 0) of a word, given by index or by string."""
    words = text.split(" ")
    if isinstance(word_place, str):
        wanted = [i for i, w in enumerate(words) if w == word_place]
    elif isinstance(word_place, int):
        wanted = [word_place]
    else:
        wanted = list(word_place)
    found = []
    if wanted:
        pieces = [tokenizer.decode([tok]).strip("#") for tok in tokenizer.encode(text)][1:-1]
        chars, w = 0, 0
        for pos, piece in enumerate(pieces):
            chars += len(piece)
            if w in wanted:
                found.append(pos + 1)
            if chars >= len(words[w]):
                w, chars = w + 1, 0
    return np.array(found)


def get_replacement_mapper_(x: str, y: str, tokenizer, max_len: int = 77) -> torch.Tensor:
    wx, wy = x.split(" "), y.split(" ")
    if len(wx) != len(wy):
        raise ValueError(
            "attention replacement edit can only be applied on prompts with the same length"
            f" but prompt A has {len(wx)} words and prompt B has {len(wy)} words."
        )
    changed = [k for k in range(len(wy)) if wy[k] != wx[k]]
    src_spans = [get_word_inds(x, k, tokenizer) for k in changed]
    tgt_spans = [get_word_inds(y, k, tokenizer) for k in changed]
    m = np.zeros((max_len, max_len))
    i = j = span = 0
    while i < max_len and j < max_len:
        if span < len(src_spans) and src_spans[span][0] == i:
            s, t = src_spans[span], tgt_spans[span]
            if len(s) == len(t):
                m[s, t] = 1
            else:
                for col in t:
                    m[s, col] = 1 / len(t)
            span += 1
            i += len(s)
            j += len(t)
        elif span < len(src_spans):
            m[i, j] = 1
            i, j = i + 1, j + 1
        else:
            m[j, j] = 1
            i, j = i + 1, j + 1
    return torch.from_numpy(m).float()


def get_replacement_mapper(prompts, tokenizer, max_len: int = 77) -> torch.Tensor:
    return torch.stack([get_replacement_mapper_(prompts[0], p, tokenizer, max_len) for p in prompts[1:]])


def get_equalizer(tokenizer, text: str, word_select, values) -> torch.Tensor:
    if isinstance(word_select, (int, str)):
        word_select = (word_select,)
    eq = torch.ones(len(values), 77)
    vals = torch.tensor(values, dtype=torch.float32)
    for w in word_select:
        for pos in get_word_inds(text, w, tokenizer):
            eq[:, pos] = vals
    return eq
