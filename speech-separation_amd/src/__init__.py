"""Drop-in shim: the reference's import paths (src.model / src.loss / src.metrics,
as used by its Hydra `_target_` strings) resolved to the MI355X-native classes."""
