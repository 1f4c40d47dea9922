// CClctrl.cpp -- see CClctrl.h.
#include "CClctrl.h"
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <vector>

CClctrl::CClctrl(int ac, char** av) : argc(ac), argv(av), argNo(1), flags(true), verbosity(2), seed(0), fileFormat(0), mode("")
{
}
bool CClctrl::isCurrentArgumentFlag() const
{
  return argNo < argc && argv[argNo][0] == '-' && argv[argNo][1] != '\0';
}
bool CClctrl::isCurrentArg(const std::string& s, const std::string& l) const
{
  return argNo < argc && (s == argv[argNo] || l == argv[argNo]);
}
std::string CClctrl::getCurrentArgument() const
{
  if(argNo >= argc) throw ndlexceptions::CommandLineError("There are not enough input parameters.");
  return argv[argNo];
}
int CClctrl::getIntFromCurrentArgument() const { return std::atoi(getCurrentArgument().c_str()); }
double CClctrl::getDoubleFromCurrentArgument() const { return std::atof(getCurrentArgument().c_str()); }
bool CClctrl::getBoolFromCurrentArgument() const
{
  const std::string a = getCurrentArgument();
  return !(a == "0" || a == "false" || a == "False" || a == "FALSE");
}
void CClctrl::unrecognisedFlag() { exitError("Unrecognised flag: " + getCurrentArgument() + " under " + getMode() + " command."); }
void CClctrl::exitError(const std::string& error)
{
  std::cerr << error << std::endl << std::endl;
  std::exit(1);
}
void CClctrl::exitNormal() { std::exit(0); }

void CClctrl::readData(CMatrix& X, CMatrix& y, const std::string fileName)
{
  if(verbosity > 1) std::cout << "Loading data." << std::endl;
  if(fileFormat != 0) exitError("Only the SVMlight file format (-f 0) is available in this build.");
  readSvmlDataFile(X, y, fileName);
  if(verbosity > 1) std::cout << "... done." << std::endl;
}
void CClctrl::readSvmlDataFile(CMatrix& X, CMatrix& y, const std::string fileName)
{
  // two passes like the reference: count rows / highest feature index, then fill.  `label idx:val ...`, 1-based
  // indices, '#' lines skipped, '\r' stripped, missing features are zero.
  std::ifstream in(fileName.c_str());
  if(!in.is_open()) throw ndlexceptions::FileReadError(fileName);
  std::vector<std::string> lines;
  std::string line;
  int maxFeat = 0;
  while(std::getline(in, line)) {
    if(!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
    if(line.empty() || line[0] == '#') continue;
    lines.push_back(line);
    std::istringstream ss(line);
    std::string tok;
    bool first = true;
    while(ss >> tok) {
      if(first) { first = false; continue; }
      const size_t c = tok.find(':');
      if(c == std::string::npos) throw ndlexceptions::FileFormatError(fileName, "feature token without ':'");
      const int f = std::atoi(tok.substr(0, c).c_str());
      if(f > maxFeat) maxFeat = f;
    }
  }
  if(verbosity > 1) {
    std::cout << "Data number of features: " << maxFeat << std::endl;
    std::cout << "Number of data: " << lines.size() << std::endl;
  }
  X.resize((unsigned int)lines.size(), (unsigned int)maxFeat);
  X.zeros();
  y.resize((unsigned int)lines.size(), 1);
  y.zeros();
  for(size_t i = 0; i < lines.size(); i++) {
    std::istringstream ss(lines[i]);
    std::string tok;
    bool first = true;
    while(ss >> tok) {
      if(first) {
        y.setVal(std::atof(tok.c_str()), (unsigned int)i);
        first = false;
        continue;
      }
      const size_t c = tok.find(':');
      const int f = std::atoi(tok.substr(0, c).c_str());
      if(f < 1 || f > maxFeat) throw ndlexceptions::FileFormatError(fileName, "feature index out of range");
      X.setVal(std::atof(tok.substr(c + 1).c_str()), (unsigned int)i, (unsigned int)(f - 1));
    }
  }
}
