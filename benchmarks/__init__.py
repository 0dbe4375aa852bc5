"""Workloads of bench.py (measurement harness, not product code).  Lives outside `pdanet_amd/` on purpose: the
`cpu_baseline` legs in here import `oracle`, and nothing under `pdanet_amd/` may."""
