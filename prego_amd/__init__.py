"""prego_amd: MI355X-native step_recognition hot path of PREGO."""
__all__ = ["config", "weights", "registry"]
