"""warpstr_amd: MI355X-native DTW-state-automaton STR caller (WarpSTR step 3)."""
__version__ = '0.1.0'
