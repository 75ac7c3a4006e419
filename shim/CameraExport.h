// CameraExport.h -- the fitted TaylorCamera model as the `mcp_camera` struct of include/mcp_ba.h (shared by shim/ChainBundle.cc and
// shim/Tracker_gpu.cc).  Needs `friend struct mcptam_hip::CameraExport;` in class TaylorCamera (shim/README.md).
#ifndef MCPTAM_HIP_CAMERA_EXPORT_H
#define MCPTAM_HIP_CAMERA_EXPORT_H

#include <mcptam/TaylorCamera.h>
#include <mcp_ba.h>
#include <stdexcept>

namespace mcptam_hip
{
// Reads the state TaylorCamera::RefreshParams leaves behind (src/TaylorCamera.cc:84-198).  Declared a friend of TaylorCamera
// because the affine matrix, rho_max, the theta normalisation and the Newton-fallback members have no public getters.
struct CameraExport
{
  static inline mcp_camera Make(TaylorCamera& cam)
  {
    mcp_camera c;
    for(int i = 0; i < 9; ++i)
      c.params[i] = cam.mv9CameraParams[i];
    c.image_size[0] = cam.mv2ImageSize[0];
    c.image_size[1] = cam.mv2ImageSize[1];
    c.affine[0] = cam.mm2Affine(0, 0);
    c.affine[1] = cam.mm2Affine(0, 1);
    c.affine[2] = cam.mm2Affine(1, 0);
    c.affine[3] = cam.mm2Affine(1, 1);
    c.center[0] = cam.mv2Center[0];
    c.center[1] = cam.mv2Center[1];
    c.min_theta = cam.mdMinTheta;
    c.max_rho = cam.mdMaxRho;
    c.theta_mean = cam.mdThetaMean;
    c.theta_std = cam.mdThetaStd;
    c.pad_ = 0;
    for(int i = 0; i < MCP_MAX_INV; ++i)
      c.inv_coeffs[i] = 0;
    if(cam.mbUsingInversePoly)
    {
      c.n_inv = cam.mvxPolyInvCoeffs.size();
      if(c.n_inv > MCP_MAX_INV)
        throw std::length_error("TaylorCamera: inverse polynomial longer than MCP_MAX_INV");
      for(int i = 0; i < c.n_inv; ++i)
        c.inv_coeffs[i] = cam.mvxPolyInvCoeffs[i];
    }
    else
    {
      // Newton fallback (src/TaylorCamera.cc:159-176): n_inv == 0, the linear inverse model seeds FindRootWithNewton on the device
      c.n_inv = 0;
      c.inv_coeffs[0] = cam.mv2LinearInvCoeffs[0];
      c.inv_coeffs[1] = cam.mv2LinearInvCoeffs[1];
    }
    return c;
  }
};
}  // namespace mcptam_hip

#endif
