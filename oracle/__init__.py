"""CPU oracle for the CSWin-UNet hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the checker / timed CPU baseline.
The product path (``cswin_unet_amd``) never imports it and fails loudly when its
HIP library is missing.

Parity status: PINNED.  ``tools/make_golden.py`` (committed) imported the real
reference (``/root/reference/networks/cswin_unet.py``) on PyTorch-CPU in the build
container and wrote ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks
every function here against those vectors.  Two things are NOT pinned by any
reference fixture and are documented as such in DESIGN.md: timm's DropPath random
stream and timm's ``trunc_normal_`` initial weights (timm is not installed and its
version is unpinned in the reference's requirements.txt).
"""
