"""picaso_amd -- MI355X (gfx950) implementation of PICASO's per-wavelength radiative-transfer
hot path behind the reference's own Python call surface.  See DESIGN.md / INTEGRATION.md."""
__version__ = "0.1.0"
