"""CPU oracle for the GAN hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain-torch (CPU, fp32) restatement of the reference's
training hot path (code/GAN/GAN_final.py, test_runs/GAN.py).  It is the
checker, never the product: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  The product package
(`mpgan_amd`, sources under `cross-modality-minipig-gan_amd/`) never imports
anything from here and fails loudly if its HIP library is missing.

Parity status (see DESIGN.md "Oracle"):
  * Discriminator (variants A and B), BCE / L1 / perceptual losses and the
    two-optimizer step are PINNED: `oracle/make_golden.py` imported the
    reference's own files in the build container (third-party modules that are
    absent there replaced by inert stubs) and wrote `tests/golden/*.npz`;
    `tests/test_oracle_golden.py` checks this restatement against them.
  * CasNetGenerator: its arithmetic lives in monai==0.4.0
    (REQUIREMENTS.txt:71), which is not vendored in the reference and not
    installed here.  The U-Net graph is restated from MONAI 0.4.0's published
    source.  PARITY UNPINNED for the generator: no
    reference test, fixture or runnable code covers it.
"""
