"""xitorch_amd — MI355X-native iterative linear algebra behind the xitorch operator API."""
__version__ = "0.1.0"
