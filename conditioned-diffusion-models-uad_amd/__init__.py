"""MI355X-native cDDPM reverse-diffusion reconstruction path (hand-written HIP behind a C ABI).

Import this package with importlib (the directory name carries a hyphen):
    pkg = importlib.import_module("conditioned-diffusion-models-uad_amd")
Submodules: synth (counter RNG + synthetic weights), schedule, engine (ctypes owner of the HIP handle),
OpenAI_Unet / cond_DDPM / DDPM_2D (host-side mirrors of the reference classes), sharding, config, build.
Nothing here imports oracle/ and nothing computes on the CPU: without the HIP library the path raises.
"""
__all__ = ["synth", "schedule", "engine", "build"]
