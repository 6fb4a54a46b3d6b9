"""Host-side counterparts of the reference's ``Helpers`` package (same public names, own implementation)."""
