"""align3d_amd — MI355X-native ICP hot path of align3d (host-side mirror of the reference API)."""
