"""CPU oracle of the self-play hot path: TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
Nothing under liuzhou_amd/ imports this package."""
