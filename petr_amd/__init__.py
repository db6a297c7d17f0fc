"""petr_amd — MI355X-native PETRHead hot path."""
__version__ = '0.1.0'
