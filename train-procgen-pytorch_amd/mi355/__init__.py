"""ctypes binding of libmi355ppo.so (include/mi355ppo.h) -- the device engine behind the drop-in
PPO / Storage / CategoricalPolicy classes."""
from .engine import Engine, EngineError, lib_path, load_library  # noqa: F401
