"""CPU oracle for the AnyV2V I2VGen-XL hot path (TEST INFRASTRUCTURE ONLY).

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker.  The product path (``anyv2v_amd``) never imports it and
fails loudly when the HIP extension is missing.

Parity status: the reference (TIGER-AI-Lab/AnyV2V) pins no results of its own
(no tests, no golden tensors).  What IS pinned here against reference code:

* the PnP hooks: ``oracle.pnp_oracle`` is checked against the reference's own
  ``i2vgen-xl/pnp_utils.py`` (imported verbatim behind ``oracle.ref_stubs``), and
  the outputs are committed as ``tests/golden/pnp_hooks_mini.pt``;
* the inverse scheduler: ``oracle.schedulers_oracle`` is checked against the
  vendored ``consisti2v/ddim_inverse_scheduler.py`` (same stub mechanism) and
  against the timestep lists / scheduler config logged in ``i2vgen-xl/demo.ipynb``.

The UNet arithmetic itself lives in third-party ``diffusers==0.26.3`` (not under
/root/reference, not installed): for that part parity is UNPINNED by the
reference; the oracle restates the published architecture and
is validated by the parameter count (1 420.5 M) and context length (145).
"""
