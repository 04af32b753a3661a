"""puflow_amd - MI355X-native discrete PU-Flow x4 upsampling path."""
__version__ = "0.1.0"
