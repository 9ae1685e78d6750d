"""Numbers the reference repository itself holds for the Chou-Suarez schemes: the tables of the NASA technical memoranda it ships as
GEOSirrad_GridComp/IrradDoc94.pdf (Chou and Suarez 1994, NASA TM 104606 vol. 3), GEOSirrad_GridComp/IrradDoc03.pdf (Chou et al.,
vol. 19) and GEOSsolar_GridComp/SolarDoc.pdf (Chou and Suarez 1999, vol. 15), transcribed by hand from the PDFs' text (data only).
They pin the restatements of irrad / sorad (oracle/chou_*_impl.h) and the GPU kernels to the reference's PUBLISHED results at the
W m-2 level - far weaker than the 1e-6 of a golden vector, but the only reference-held numbers there are for these routines (their
sources need MAPL and cannot be built here).

IrradDoc94, section 7.9 "Sample Program" / 7.10 "Verification Output of Sample Program" (pp. 75-81): a 75-layer mid-latitude
summer atmosphere (McClatchey et al. 1972; ICRCCM conventions: CO2 300 ppmv, specific humidity 4e-6 above the tropopause, dp = 23.775
mb below 100 mb, dlog10 p = 0.15 above), ts = 294 K, black surface, and the net downward fluxes the 1994 code returned for it with
the HIGH option (its clear-sky column `flc` is used here; the all-sky column belongs to the 1994 cloud input - optical thickness and
cover per layer - which today's irrad replaced by water contents and effective radii).
"""
import numpy as np

# level pressures (mb), top down, 76 levels
PL_MB = np.array([
    0.0, .0006244, .0008759, .001229, .001723, .002417, .003391, .004757, .006672, .009359, .01313, .01842, .02583, .03623, .05082,
    .07129, 0.10, 0.14, 0.20, 0.28, 0.39, 0.54, 0.76, 1.07, 1.50, 2.10, 2.95, 4.14, 5.80, 8.14, 11.42, 16.01, 22.46, 31.51, 44.20, 62.00,
    85.78, 109.55, 133.32, 157.10, 180.88, 204.65, 228.43, 252.20, 275.98, 299.75, 323.52, 347.30, 371.08, 394.85, 418.63, 442.40,
    466.17, 489.95, 513.72, 537.50, 561.28, 585.05, 608.83, 632.60, 656.38, 680.15, 703.92, 727.70, 751.47, 775.25, 799.03, 822.80,
    846.58, 870.35, 894.13, 917.90, 941.67, 965.45, 989.22, 1013.00])
# layer temperatures (K), 75 layers
TA_K = np.array([
    209.86, 210.20, 210.73, 211.27, 211.81, 212.35, 212.89, 213.44, 213.98, 214.53, 215.08, 215.62, 216.17, 216.74, 218.11, 223.20,
    230.04, 237.14, 244.46, 252.00, 259.76, 267.70, 274.93, 274.60, 269.38, 262.94, 256.45, 250.12, 244.31, 238.96, 233.74, 228.69,
    224.59, 221.75, 219.10, 216.64, 215.76, 215.75, 215.78, 216.22, 219.15, 223.79, 228.29, 232.45, 236.33, 239.92, 243.32, 246.53,
    249.56, 252.43, 255.14, 257.69, 260.11, 262.39, 264.57, 266.66, 268.67, 270.60, 272.48, 274.29, 276.05, 277.75, 279.41, 281.02,
    282.59, 284.09, 285.53, 286.86, 288.06, 289.13, 290.11, 291.03, 291.91, 292.76, 293.59])
# layer specific humidity (g/g)
WA = np.array([0.400e-05] * 38 + [
    0.406e-05, 0.520e-05, 0.115e-04, 0.275e-04, 0.572e-04, 0.107e-03, 0.166e-03, 0.223e-03, 0.285e-03, 0.360e-03, 0.446e-03, 0.547e-03,
    0.655e-03, 0.767e-03, 0.890e-03, 0.103e-02, 0.118e-02, 0.136e-02, 0.159e-02, 0.190e-02, 0.225e-02, 0.264e-02, 0.306e-02, 0.351e-02,
    0.399e-02, 0.450e-02, 0.504e-02, 0.560e-02, 0.619e-02, 0.680e-02, 0.742e-02, 0.805e-02, 0.869e-02, 0.935e-02, 0.100e-01, 0.107e-01,
    0.113e-01])
# layer ozone mass mixing ratio (g/g)
OA = np.array([
    0.6427e-07, 0.2022e-06, 0.2458e-06, 0.2896e-06, 0.3337e-06, 0.3779e-06, 0.4224e-06, 0.4671e-06, 0.5120e-06, 0.5572e-06, 0.6026e-06,
    0.6482e-06, 0.6940e-06, 0.7401e-06, 0.7934e-06, 0.1009e-05, 0.1313e-05, 0.1635e-05, 0.1976e-05, 0.2336e-05, 0.2716e-05, 0.3117e-05,
    0.3590e-05, 0.4645e-05, 0.5897e-05, 0.7649e-05, 0.9102e-05, 0.9598e-05, 0.9944e-05, 0.1008e-04, 0.9900e-05, 0.8526e-05, 0.7098e-05,
    0.5761e-05, 0.4231e-05, 0.2604e-05, 0.1523e-05, 0.1028e-05, 0.7859e-06, 0.5983e-06, 0.4483e-06, 0.3818e-06, 0.3021e-06, 0.2516e-06,
    0.2117e-06, 0.1932e-06, 0.1761e-06, 0.1599e-06, 0.1466e-06, 0.1366e-06, 0.1270e-06, 0.1180e-06, 0.1094e-06, 0.1028e-06, 0.9748e-07,
    0.9238e-07, 0.8828e-07, 0.8458e-07, 0.8098e-07, 0.7782e-07, 0.7490e-07, 0.7205e-07, 0.6939e-07, 0.6706e-07, 0.6480e-07, 0.6258e-07,
    0.6069e-07, 0.5928e-07, 0.5789e-07, 0.5651e-07, 0.5517e-07, 0.5398e-07, 0.5281e-07, 0.5167e-07, 0.5054e-07])
TS_K, CO2_LW = 294.0, 300.0e-6
# IrradDoc94 7.10: clear-sky net downward flux `flc` (W m-2) at the 76 levels, HIGH option
FLC_1994 = np.array([
    -293.07, -293.07, -293.07, -293.06, -293.06, -293.06, -293.06, -293.05, -293.05, -293.05, -293.05, -293.05, -293.04, -293.04, -293.03,
    -293.03, -293.03, -293.02, -293.00, -292.96, -292.89, -292.77, -292.54, -292.10, -291.51, -290.87, -290.16, -289.39, -288.48,
    -287.44, -286.30, -285.04, -283.52, -282.17, -280.86, -279.08, -278.09, -277.40, -276.75, -276.39, -276.51, -275.53, -272.66,
    -268.06, -262.55, -256.92, -251.64, -246.54, -241.45, -236.35, -231.24, -226.19, -221.29, -216.44, -211.61, -206.82, -201.99,
    -196.97, -191.82, -186.36, -180.95, -175.29, -169.63, -163.82, -157.89, -151.75, -145.49, -139.01, -132.32, -125.49, -118.64,
    -111.85, -104.80, -97.57, -90.04, -81.15])
ST4_1994 = -423.62          # upward surface emission the memo prints (sigma ts^4)
# IrradDoc94 Table 8 (p. 27) and IrradDoc03 Table 14 (p. 48): clear mid-latitude summer, downward flux at the surface / upward flux at the
# top of the atmosphere (W m-2), parameterization (HIGH) and line-by-line; H2O, CO2 (300 ppmv), O3 only
MLS_SFC_DOWN = {"param_1994": 342.45, "lbl_1994": 339.93, "param_2003": 342.37, "lbl_2003": 339.59}
MLS_TOA_UP = {"param_1994": 293.07, "lbl_1994": 293.10, "param_2003": 293.03, "lbl_2003": 293.99}
# IrradDoc03 Table 16 (p. 50): effect of the minor absorption bands (CH4, N2O, CFCs, minor CO2 bands) on the mid-latitude summer fluxes
MLS_MINOR_BANDS = {"toa_up": -4.91, "sfc_down": +2.68}

# SolarDoc.pdf, section 8 (pp. 31-35): mid-latitude summer atmosphere, CO2 350 ppmv, solar zenith angle 60 deg, surface albedo 0.2.
# Table 8 (p. 33): absorption by H2O, O3, CO2, O2 only, scattering EXCLUDED (a configuration sorad cannot be put into: Rayleigh
# scattering is built in): net downward flux at TOA / surface and atmospheric absorption, parameterization (high spectral resolution)
SW_CLEAR_NO_SCATTERING = {"toa": (579.1, 581.5), "sfc": (431.7, 433.3), "atm": (147.4, 148.2)}
# Table 9 (p. 35): scattering by gases and cloud included; a stratus deck of visible optical thickness 9.7 between 800 and 920 hPa
# (5 sub-layers of 24 hPa, 14.9 g m-2 of liquid water each, effective radius 12 um, overcast): parameterization (detailed)
SW_STRATUS = {"toa": (346.4, 354.9), "sfc": (186.9, 187.2), "atm": (159.6, 167.7)}
SW_COSZ, SW_ALBEDO, SW_CO2, SW_CLOUD_LWP_PER_LAYER, SW_CLOUD_REFF = 0.5, 0.2, 350.0e-6, 14.9, 12.0
SW_CLOUD_LAYERS = range(66, 71)      # 0-based layers between the levels 799.03 and 917.90 mb of PL_MB

# IrradDoc94 Table 9 (p. 40, HIGH option) / IrradDoc03 Table 14 (p. 48): per spectral band, clear mid-latitude summer:
# (downward flux at the surface, upward flux at the top of the atmosphere), W m-2
BANDS_1994 = {"0-340": (50.97, 34.04), "340-540": (81.23, 60.01), "540-800": (107.58, 68.40), "800-980": (28.35, 58.50),
              "980-1100": (12.86, 21.81), "1100-1380": (27.95, 38.21), "1380-1900": (30.35, 7.22), "1900-3000": (3.16, 4.88)}
BANDS_2003 = [(51.04, 34.40), (81.23, 60.54), (107.29, 68.29), (27.01, 58.72), (12.54, 21.41), (9.63, 21.42), (19.78, 15.67), (30.68, 7.61),
              (3.17, 5.00)]            # irrad's bands 1..9 (0-340, 340-540, 540-800, 800-980, 980-1100, 1100-1215, 1215-1380, 1380-1900, 1900-3000)
# water-vapour continuum coefficients xke (cm2 g-1) of irrad's bands 1..9: the reference builds with the CKD 2.3 set and carries Roberts et
# al.'s - the set both memoranda's tables were computed with - as a comment (GEOSirrad_GridComp/irradconstants.F90:25-31); the 1994
# code had no continuum below 540 cm-1 (IrradDoc94 p. 28), i.e. none in band 2
XKE_ROBERTS = np.array([0.00, 339.00, 27.40, 15.8, 9.40, 7.75, 7.70, 0.0, 0.0])
XKE_1994 = np.array([0.00, 0.00, 27.40, 15.8, 9.40, 7.75, 7.70, 0.0, 0.0])


def irrad_inputs(m=2):
    """the sample atmosphere as inputs of today's irrad (synth.chou_lw_inputs conventions: numpy [row][column]): clear sky, no aerosol,
    one black surface type at ts"""
    npl = TA_K.size
    f = lambda a: np.repeat(np.asarray(a, dtype=np.float64)[:, None], m, axis=1)
    zero = np.zeros((npl, m))
    return dict(ple=f(PL_MB * 100.0), ta=f(TA_K), wa=f(WA), oa=f(OA), tb=np.full(m, TS_K), co2=CO2_LW, n2o=zero, ch4=zero, cfc11=zero, cfc12=zero,
                cfc22=zero, fcld=zero, cwc=np.zeros((4, npl, m)), reff=np.full((4, npl, m), 10.0), ict=int(np.sum(PL_MB < 400.0)),
                icb=int(np.sum(PL_MB < 700.0)), ns=1, fs=np.ones((1, m)), tg=np.full((1, m), TS_K), tv=np.full((1, m), TS_K), eg=np.ones((10, 1, m)),
                ev=np.zeros((10, 1, m)), rv=np.zeros((10, 1, m)), na=0, nb=10, taua=np.zeros((10, npl, m)), ssaa=np.zeros((10, npl, m)),
                asya=np.zeros((10, npl, m)))


def sorad_inputs(hk_uv, hk_ir, cloud, m=2, grav=9.80665):
    """SolarDoc section 8's cases on the same 75-layer mid-latitude summer atmosphere (its 5 cloud sub-layers of 24 hPa between 800 and 920
    hPa are layers 67..71 of this grid: 799.03 .. 917.90 mb) as inputs of today's sorad (synth.chou_sw_inputs conventions)"""
    npl = TA_K.size
    f = lambda a: np.repeat(np.asarray(a, dtype=np.float64)[:, None], m, axis=1)
    cwc = np.zeros((4, npl, m)); fcld = np.zeros((npl, m))
    if cloud:
        for k in SW_CLOUD_LAYERS:
            cwc[1, k, :] = SW_CLOUD_LWP_PER_LAYER * 1e-3 * grav / ((PL_MB[k + 1] - PL_MB[k]) * 100.0)       # kg/kg from g m-2
            fcld[k, :] = 1.0
    reff = np.zeros((4, npl, m)); reff[0] = 40.0; reff[1] = SW_CLOUD_REFF; reff[2] = 100.0; reff[3] = 140.0
    alb = np.full(m, SW_ALBEDO)
    return dict(cosz=np.full(m, SW_COSZ), pl=f(PL_MB), ta=f(TA_K), wa=f(WA), oa=f(OA), co2=SW_CO2, cwc=cwc, fcld=fcld,
                ict=int(np.sum(PL_MB < 400.0)), icb=int(np.sum(PL_MB < 700.0)), reff=reff, hk_uv=np.asarray(hk_uv, dtype=np.float64),
                hk_ir=np.asarray(hk_ir, dtype=np.float64), rsuvbm=alb, rsuvdf=alb, rsirbm=alb, rsirdf=alb, nb=8, taua=np.zeros((8, npl, m)),
                ssaa=np.zeros((8, npl, m)), asya=np.zeros((8, npl, m)))


SW_S0 = 1367.0      # W m-2: the memorandum does not state its solar constant; 1367 makes its Table 8 consistent (insolation 683.5 W m-2)
