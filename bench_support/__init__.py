"""Parts of the benchmark (`bench.py` = the headline line; `run_legs.py` = the secondary legs, written to a detail file).
Measurement code only - nothing here is imported by the product (`mono_lidar_depth_amd/`)."""
