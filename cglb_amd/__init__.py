"""MI355X-native CGLB quadratic-term solver (hand-written HIP behind a C ABI).

Public surface mirrors the reference's `cglb.backend` seams; see DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"
