"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A CPU (PyTorch fp32, eager) restatement of the reference's algorithm for the h-Edit sampling
hot path: scheduler algebra, the four h-Edit loops, DDPM inversion, the
Prompt-to-Prompt controller / LocalBlend / attention processor, and the SD-1.x eps-network the
loops call.  Every function cites the reference file:line it follows.

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg -- as the checker / the timed CPU baseline only.  The product package
(``h-edit_amd/hedit``) never imports it and fails loudly when its HIP library is missing.

Pinning status
  * loops, scheduler algebra, controller, LocalBlend, processor: PINNED against golden vectors
    produced by running the reference's own modules (tests/golden/make_golden.py, committed
    fixtures tests/golden/g1..g7); the text + style loop of text-guided-n-style
    (loops.h_edit_p2p_implicit_style) likewise on g9, reproduced bit for bit.
  * MasaCtrl (oracle/masactrl.py + loops.h_edit_masactrl_implicit): PINNED on vectors from running the
    reference's MutualSelfAttentionControl / registration / h_Edit_masactrl_implicit (g13).
  * Plug-and-Play (oracle/pnp.py + loops.h_edit_pnp_implicit): PINNED on vectors from running the reference's
    pnp_utils hooks and pnp_h_edit loop patched into the oracle SD UNet (g14).
  * face-swapping path: the pixel DDPM UNet (oracle/ddpm_unet.py) and the SDE inversion / h_Edit_R face loop
    (oracle/face_loops.py) are PINNED on vectors from running the reference's own in-tree
    face-swapping/diffusion/diffusion.py, inversion/sde_inversion.py and inversion/h_edit_R.py (g11).
  * SD-1.x UNet arithmetic (oracle/sd_unet.py): the reference delegates it to the third-party
    package diffusers==0.18.0 (text-guided/environment_p2p.yaml:88), which is absent from
    /root/reference and from this image, and the reference holds no test or golden vector for
    it => PARITY UNPINNED for the network body.  What is pinned: the attention op order (through
    the reference's P2PCrossAttnProcessor, g6), and the parameter inventory (859.5 M params,
    diffusers state_dict key names) -- see DESIGN.md §Oracle.
  * SD-1.x image autoencoder (oracle/sd_vae.py): same situation -- diffusers' AutoencoderKL, no
    reference test or vector => PARITY UNPINNED for the body; pinned: the parameter inventory
    (248 tensors, 83,653,863 parameters, diffusers key names).  The host image I/O around it
    (load_512) is not oracle code at all: the product function is checked bit-exactly against
    vectors produced by the reference's own load_512 (tests/golden/g8_load512.npz).
"""
