"""`AutoModel(..., trust_remote_code=True, remote_code=".../funasr_amd/remote_code.py")`: FunASR imports this file
(funasr/utils/dynamic_import.py:23-45) before it looks the model up, which re-points the registry at the HIP classes."""
from funasr_amd.install import install

install()
