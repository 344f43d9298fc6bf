"""lia_amd -- MI355X-native implementation of LIA's weight-offloaded cooperative decoder hot path.

Host-side mirror of the reference's interface (run.py flags -> generate() -> greedy loop ->
OPTDecoder.forward scheduler -> decoder_layer operator) over liblia_hip.so (include/lia_hip.h).
"""
__version__ = "0.1.0"
