"""What a plain device-to-device copy sustains on this box (context for the sort's roofline fraction):
bytes read + bytes written per second for torch's copy kernel and for hipMemcpyAsync D2D."""
import torch, time
dev = torch.device("cuda", 0)
for gb in (1, 4):
    n = gb * (1 << 30)
    a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
    a.random_(0, 255)
    for name, fn in (("tensor.copy_", lambda: b.copy_(a)), ("int4 view copy_", lambda: b.view(torch.int32).copy_(a.view(torch.int32)))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%-16s %d GiB: %.3f ms  read+write %.2f TB/s" % (name, gb, ms, 2 * n / ms / 1e9))
    s = torch.empty(n // 4, dtype=torch.int32, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    x = a.view(torch.int32).sum(); torch.cuda.synchronize()
    e0.record()
    for _ in range(10): x = a.view(torch.int32).sum()
    e1.record(); torch.cuda.synchronize()
    print("sum (read only)  %d GiB: %.3f ms  read %.2f TB/s" % (gb, e0.elapsed_time(e1) / 10, n / (e0.elapsed_time(e1) / 10) / 1e9))
    e0.record()
    for _ in range(10): s.fill_(7)
    e1.record(); torch.cuda.synchronize()
    print("fill (write only) %d GiB: %.3f ms  write %.2f TB/s" % (gb, e0.elapsed_time(e1) / 10, n / (e0.elapsed_time(e1) / 10) / 1e9))
