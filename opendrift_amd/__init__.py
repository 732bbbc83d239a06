"""opendrift_amd -- MI355X-native particle-advection hot path behind OpenDrift's model/reader API."""
__version__ = '0.1.0'
