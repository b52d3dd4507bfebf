"""MI355X-native BASD distillation train-step path.

Sub-packages mirror the reference's ``src/`` layout for the hot path only:
``losses`` (layer_selector, relational, combined), ``models`` (teacher, vit),
``training`` (trainer, optimizer, data-parallel), plus ``_native`` (ctypes
binding of the C-ABI HIP library built from ``csrc/``).
"""
__version__ = "0.1.0"
