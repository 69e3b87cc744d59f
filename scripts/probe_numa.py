import os, torch, glob
p = torch.cuda.get_device_properties(0)
print({k: getattr(p, k) for k in dir(p) if k.startswith("pci")})
bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
for f in ("numa_node", "local_cpulist"):
    try: print(f, open(f"/sys/bus/pci/devices/{bdf}/{f}").read().strip())
    except Exception as e: print(f, "ERR", e)
print("affinity", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], "...", "cpu now", os.sched_getcpu() if hasattr(os, "sched_getcpu") else None)
for n in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")): print(n, open(n).read().strip())
print(os.popen("lscpu | grep -i -E 'numa|model name|socket|thread'").read())
print(os.popen("cat /sys/class/kfd/kfd/topology/nodes/*/properties 2>/dev/null | grep -E 'cpu_cores_count|simd_count|location_id|domain' | head -40").read())
