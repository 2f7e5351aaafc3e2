import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
def cdbg():
    import bcalm_amd
    g = bcalm_amd.Graph(31, 2); g.generate_reads(1000, 150, 3); g.run(); n = g.stats()["n_unitigs"]; g.close(); return n
def tor():
    import torch
    return float(torch.zeros(4, device="cuda").sum().item())
try:
    if mode == "cdbg_torch": print("cdbg", cdbg()); print("torch", tor())
    elif mode == "torch_cdbg": print("torch", tor()); print("cdbg", cdbg())
    elif mode == "import_torch_cdbg_torch": import torch; print("cdbg", cdbg()); print("torch", tor())
    elif mode == "load_cdbg_torch_cdbg":
        import bcalm_amd; bcalm_amd.load(); print("torch", tor()); print("cdbg", cdbg())
except Exception as e:
    print("FAIL", mode, type(e).__name__, str(e)[:100])
