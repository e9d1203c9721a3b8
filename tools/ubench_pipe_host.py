import ctypes, sys, os
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.init(); x = torch.zeros(10, device="cuda"); torch.cuda.synchronize()
    print("torch initialised")
if len(sys.argv) > 2 and sys.argv[2] == "afg":
    sys.path.insert(0, "audio-formats_amd")
    import afgpu
    afgpu.lib()
    print("afg loaded")
L = ctypes.CDLL(os.path.abspath("tools/ubench_pipe.so"))
L.ubench_pipe_run()
