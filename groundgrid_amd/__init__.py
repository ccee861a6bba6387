"""groundgrid_amd -- MI355X-native GroundGrid hot path."""
