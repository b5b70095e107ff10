"""Drop-in for the reference's ``memento`` package API (memento/__init__.py:1), HIP-backed."""
