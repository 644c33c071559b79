"""MI355X-native per-frame surfel-fusion hot path."""
