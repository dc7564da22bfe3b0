"""MI355X-native (gfx950) inference path for FunASR's Paraformer / SenseVoice models.

Host-side mirror of the reference's registry modules on top of the C ABI in include/paraformer_hip.h; see
DESIGN.md and INTEGRATION.md. Importing the package does not require a GPU; running any module does.
"""
__version__ = "0.1.0"
