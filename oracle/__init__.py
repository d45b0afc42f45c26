"""TEST INFRASTRUCTURE — not product code.

CPU (PyTorch fp32) restatement of the reference's EGM-UNet hot path, used only
as the parity checker by ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg.  Nothing under ``egm_unet_amd/`` may
import this package; the product path runs on the HIP library only and fails
loudly when that library is missing.

Pinning: the reference ships no tests or golden vectors of its own (SURVEY.md
§4), so this restatement is pinned by fixtures captured from the reference
itself, imported by file path in the build container by
``tools/make_golden.py`` and committed under ``tests/golden/``.
``tests/test_oracle_golden.py`` checks every function here against them.
"""
