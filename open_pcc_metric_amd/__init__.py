"""MI355X-native engine for open-pcc-metric's CloudPair / MetricCalculator hot path."""
__version__ = "0.1.0"
