"""a3t_amd: MI355X-native (gfx950 HIP) implementation of the A3T masked-mel training step and
ParallelWaveGAN inference path behind the reference's ESPnet2 plugin surface."""
from .config import A3TConfig, config_c2  # noqa: F401

__version__ = "0.1.0"
