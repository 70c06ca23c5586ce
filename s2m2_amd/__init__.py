"""MI355X-native (gfx950) hot path of the S2M2 stereo matcher: hand-written HIP kernels behind a C ABI
(``include/s2m2_hip.h`` -> ``s2m2_amd/lib/libs2m2_hip.so``) and the Python mirror of the reference module interface.

    from s2m2_amd.model import S2M2, build_model      # drop-in for s2m2.core.model.s2m2.S2M2
    from s2m2_amd.utils import load_model, image_pad, image_crop, run_stereo_matching

Nothing here falls back to the CPU or to PyTorch compute: the kernel library must be built (``python -m s2m2_amd.build``).
"""
