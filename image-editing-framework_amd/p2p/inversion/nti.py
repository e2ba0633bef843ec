"""Null-text inversion (`/root/reference/p2p/inversion/nti.py:9-45`).

Per timestep the reference Adam-optimises the unconditional embedding through the UNet, which
needs the UNet's BACKWARD with respect to `encoder_hidden_states` (cross-attention K/V
projections -> attention -> everything downstream).  The activation-gradient kernels are the
first "next" row of the scope table and are not built yet; this class keeps
the reference's name and signature and fails loudly instead of silently running a slow or wrong
substitute.  The numerics of the loop itself are pinned in the oracle against the reference
(`oracle/p2p_ref.py:null_optimization`, fixture G8).  `P2P_NTI.text2image_ldm_stable` already accepts
a precomputed `uncond_embeddings_list` (e.g. loaded by `dataset.pie.PIE_NTI_Inversion`).
"""
from .ddim import ddim_inversion


class NTI(ddim_inversion):
    def null_optimization(self, model, latents, context, num_inner_steps, epsilon, guidance_scale):
        raise NotImplementedError(
            "null-text optimisation needs the UNet backward w.r.t. encoder_hidden_states, which is not built yet "
            "(DESIGN.md, 'What comes next' #1).  Use --inversion_type ddim, or pass precomputed "
            "uncond_embeddings_list to P2P_NTI.text2image_ldm_stable.")
