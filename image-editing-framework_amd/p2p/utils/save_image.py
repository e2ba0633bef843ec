"""`save_img` / `save_images` (`/root/reference/p2p/utils/save_image.py:6-31`): PNG writers."""
import numpy as np
from PIL import Image


def save_img(image, path):
    """image: uint8 HWC (or 1xHWC) array -> PNG"""
    arr = np.asarray(image)
    if arr.ndim == 4:
        arr = arr[0]
    Image.fromarray(arr.astype(np.uint8)).save(path)


class PngWriter:
    """PNG encoding (zlib, ~30 ms per 512x512 image) on host threads, so the GPU loop of a PIE driver never waits for a
    file between two edits (the reference writes synchronously, `/root/reference/p2p/utils/save_image.py:6-14`,
    `p2p/test.py:171-178`).  A writer's exception (disk full, bad path) surfaces at the NEXT `save_*` call after the file
    failed -- not only at the end of the dataset -- and at `flush()` / `close()`, which wait for every file; used as a context
    manager the pool is drained even when the GPU loop itself raises."""

    def __init__(self, workers: int = 2):
        from concurrent.futures import ThreadPoolExecutor
        self._pool = ThreadPoolExecutor(max_workers=workers)
        self._pending = []

    def _reap(self):
        """drop finished writes, re-raising the first one that failed"""
        still = []
        for f in self._pending:
            if f.done():
                f.result()
            else:
                still.append(f)
        self._pending = still

    def save_img(self, image, path):
        self._reap()
        self._pending.append(self._pool.submit(save_img, np.array(image, copy=True), path))

    def save_pil(self, image, path):
        self._reap()
        self._pending.append(self._pool.submit(image.save, path))

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        try:
            self.close()
        except Exception:
            if exc_type is None:
                raise                   # a write failed and nothing else did: report it
        return False                    # otherwise the loop's own exception propagates (the pool is shut down either way)

    def flush(self):
        pending, self._pending = self._pending, []
        for f in pending:
            f.result()

    def close(self):
        try:
            self.flush()
        finally:
            self._pool.shutdown()


def save_images(images, path, num_rows=1, offset_ratio=0.02):
    """uint8 images [N,H,W,3] tiled into one PNG grid with white gutters"""
    images = [np.asarray(im).astype(np.uint8) for im in (images if not isinstance(images, np.ndarray) or images.ndim == 4 else [images])]
    n = len(images)
    pad = (-n) % num_rows
    h, w, c = images[0].shape
    images += [np.full((h, w, c), 255, np.uint8)] * pad
    cols = len(images) // num_rows
    off = int(h * offset_ratio)
    grid = np.full((h * num_rows + off * (num_rows - 1), w * cols + off * (cols - 1), c), 255, np.uint8)
    for i, im in enumerate(images):
        r, q = divmod(i, cols)
        grid[r * (h + off): r * (h + off) + h, q * (w + off): q * (w + off) + w] = im
    Image.fromarray(grid).save(path)
