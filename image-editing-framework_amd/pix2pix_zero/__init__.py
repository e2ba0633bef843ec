"""Pix2Pix-zero method folder (`/root/reference/pix2pix-zero/`; the hyphen is not importable, hence the underscore)."""
