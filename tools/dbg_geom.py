import sys; sys.path.insert(0,'.')
import numpy as np, kifs_raymarching_amd as K
gs = K.GraphicState(0)
gs.set_camera(K.CameraData(origin_distance=3.0)); gs.update_options(K.GuiData(primitive_shape=K.PrimitiveShape.Torus))
for i,(w,h) in enumerate([(32,16),(1,1),(33,9),(31,7),(100,3),(65,130),(200,135),(160,90),(96,54),(1920,1080),(4096,4096),(480,270),(256,256)]):
    gs.update_screen_data(K.ScreenData(w,h))
    try:
        img = gs.render(); print(i,(w,h),"ok", img.shape)
        img = gs.render(y0=0,y1=max(1,h//2)); print(i,(w,h),"band ok")
    except Exception as e:
        print(i,(w,h),"FAIL",e)
