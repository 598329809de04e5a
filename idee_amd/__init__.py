"""idee_amd — MI355X-native (gfx950) implementation of the IDEE encoder -> codebook -> classifier
hot path behind the reference's models/build.py module API.  See DESIGN.md."""
__version__ = "0.1.0"
