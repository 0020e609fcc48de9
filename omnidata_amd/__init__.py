"""omnidata_amd: MI355X-native DPT-Hybrid-384 depth / surface-normal inference path."""
__version__ = "0.1.0"
