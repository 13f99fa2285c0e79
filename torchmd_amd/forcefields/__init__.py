from .forcefield import ForceField, ForceFieldBase
from .ff_yaml import YamlForceField
from .ff_prmtop import PrmtopForceField

__all__ = ["ForceField", "ForceFieldBase", "YamlForceField", "PrmtopForceField"]
