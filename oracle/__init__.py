"""CPU oracle for the quantized-DiT denoising path.

TEST INFRASTRUCTURE ONLY.  This package restates, in plain torch-CPU fp32
arithmetic, what the reference (thu-nics/ViDiT-Q, ``/root/reference``) computes
on its fake-quant path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package
(``vidit-q_amd``) never does and fails loudly when its HIP library is missing.

Modules: ``fakequant`` (quantizer / QuantLayer arithmetic), ``stdit_ref`` (STDiT block + model + DDIM/CFG step),
``pixart_ref`` (PixArt-MS block + model), ``ref_import`` (authoring-container helper that imports the reference with
stubs; never used on the GPU box).  The t2i DPM-Solver++ driver is host logic of the product
(``vidit-q_amd/t2i/dpm_solver.py``) and is pinned by a trajectory of the reference's solver in the CPU tests.

Parity status: the reference ships no tests and no golden vectors
(SURVEY.md §4), so the oracle is pinned against outputs of the reference
itself, generated in the authoring container by importing it with stubs for
its absent third-party imports (``oracle/ref_import.py``; vectors under
``tests/golden/``, generator ``tests/golden/make_golden.py``).  Attention
numerics at the flash-attn / xformers boundary are restated from those
libraries' published semantics (softmax(q k^T * d^-1/2) v, block-diagonal
mask) and are "parity unpinned" by any reference test.
"""
