import sys, os
sys.path.insert(0, os.getcwd())
which = sys.argv[1]
if which == "A":
    import torch
    print("A avail", torch.cuda.is_available()); torch.cuda.init(); print("A init ok")
    from edge_alignment_amd import capi
    print("A count", capi.device_count())
    P = capi.Problem(525, 525, 319.5, 239.5); print("A problem ok")
    x = torch.zeros(4, device="cuda"); print("A tensor ok")
else:
    from edge_alignment_amd import capi
    print("B count", capi.device_count())
    P = capi.Problem(525, 525, 319.5, 239.5); print("B problem ok")
    import torch
    print("B avail", torch.cuda.is_available())
    try:
        torch.cuda.init(); print("B init ok")
        x = torch.zeros(4, device="cuda"); print("B tensor ok")
    except Exception as e:
        print("B fail", repr(e)[:200])
