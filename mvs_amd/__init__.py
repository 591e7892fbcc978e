"""mvs_amd -- MI355X-native MVSNet cost-volume path."""
__version__ = "0.1.0"
