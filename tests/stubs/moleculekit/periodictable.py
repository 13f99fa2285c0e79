"""`periodictable_by_number[z].mass` as `torchmd/npzmol.py:2,19` uses it (imported by run.py at module level)."""
from types import SimpleNamespace

_MASS = {1: 1.008, 6: 12.011, 7: 14.007, 8: 15.999, 9: 18.998, 11: 22.990, 12: 24.305, 15: 30.974, 16: 32.06, 17: 35.45,
         18: 39.948, 19: 39.098, 20: 40.078}
periodictable_by_number = {z: SimpleNamespace(mass=m, number=z) for z, m in _MASS.items()}
