"""MI355X-native sparse-GP inference core for edr-gp (drop-in for edrgp.gp_model)."""
__version__ = "0.1.0"
