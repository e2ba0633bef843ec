"""PIE-Bench dataset views (`/root/reference/p2p/dataset/pie.py:8-51`) + a synthetic stand-in.

`PIE[i] -> (image_path, source_prompt, target_prompt)` with the `[`/`]` markers stripped from the
prompts (:19-23), filtered by editing category.  The benchmark download is not available here
(SURVEY.md §2.1 row 12), so `SyntheticPIE` yields the same triple shape over generated images and
the reference's default prompt pairs; `test.py --synthetic N` uses it to measure images/sec.
"""
import json
import os

import numpy as np
import torch
from PIL import Image


class PIE(torch.utils.data.Dataset):
    def __init__(self, dataset_path, inversion_path=None, category=0):
        with open(os.path.join(dataset_path, "mapping_file.json")) as f:
            mapping = json.load(f)
        self.items = []
        for key, e in mapping.items():
            if int(e["editing_type_id"]) != int(category):
                continue
            self.items.append((os.path.join(dataset_path, "annotation_images", e["image_path"]),
                               e["original_prompt"].replace("[", "").replace("]", ""),
                               e["editing_prompt"].replace("[", "").replace("]", "")))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, idx):
        return self.items[idx]


class PIE_NTI_Inversion(PIE):
    """adds the precomputed inversion latent + null-text embeddings stored next to each image (:25-51)"""

    def __init__(self, dataset_path, inversion_path, category=0):
        super().__init__(dataset_path, inversion_path, category)
        self.dataset_path, self.inversion_path = dataset_path, inversion_path

    def __getitem__(self, idx):
        path, src, tgt = self.items[idx]
        rel = os.path.relpath(path.split(".")[0], os.path.join(self.dataset_path, "annotation_images"))
        d = os.path.join(self.inversion_path, rel)
        latent = torch.load(os.path.join(d, "inversion_latent.pt"), map_location="cpu")
        uncond = torch.load(os.path.join(d, "uncond_embeddings_list.pt"), map_location="cpu")
        return path, src, tgt, latent, uncond


_SYN_PAIRS = [
    ("a gray horse in the field", "a whie horse in the field"),                      # edit_real.py defaults (equal length -> replace)
    ("a photo of a house on a mountain", "a photo of a house on a mountain at fall"),  # edit_syn.py defaults (refine)
    ("a cat sitting on a bench", "a dog sitting on a bench"),
    ("a bowl of fruit", "a bowl of strawberries and fruit on the table"),
]


class SyntheticPIE(torch.utils.data.Dataset):
    """n seeded 512x512 RGB images written under `root` + cycling prompt pairs"""

    def __init__(self, root, n, size=512, seed=0):
        os.makedirs(root, exist_ok=True)
        self.items = []
        for i in range(n):
            path = os.path.join(root, f"syn_{i:04d}.png")
            if not os.path.exists(path):
                # pixels depend on (seed, i) only — not on which files already exist — and the file appears atomically:
                # under torchrun every rank builds this dataset into the same directory
                rng = np.random.RandomState([seed, i])
                low = rng.randint(0, 256, size=(size // 32, size // 32, 3)).astype(np.uint8)
                tmp = f"{path}.{os.getpid()}.tmp"
                Image.fromarray(low).resize((size, size), Image.BICUBIC).save(tmp, format="PNG")
                os.replace(tmp, path)
            src, tgt = _SYN_PAIRS[i % len(_SYN_PAIRS)]
            self.items.append((path, src, tgt))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, idx):
        return self.items[idx]
