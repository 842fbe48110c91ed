import amdsmi, json
amdsmi.amdsmi_init()
h=amdsmi.amdsmi_get_processor_handles()[0]
def plain(v,d=0):
    if isinstance(v,(int,float,str,bool)) or v is None: return v
    if isinstance(v,dict): return {str(k):plain(x,d+1) for k,x in v.items()}
    if isinstance(v,(list,tuple)): return [plain(x,d+1) for x in v[:8]]
    return str(v)
for fn in ("amdsmi_get_violation_status","amdsmi_get_gpu_metrics_info","amdsmi_get_power_cap_info","amdsmi_get_gpu_metrics_header_info"):
    try:
        print(fn, json.dumps(plain(getattr(amdsmi,fn)(h)))[:6000])
    except Exception as e:
        print(fn,'ERR',e)
print([n for n in dir(amdsmi) if 'thrott' in n.lower() or 'violation' in n.lower() or 'edc' in n.lower() or 'tdc' in n.lower() or 'current' in n.lower() or 'limit' in n.lower()])
